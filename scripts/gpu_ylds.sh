#!/bin/bash
OUT=gpurun_out/${1:-fc}
mkdir -p $OUT
for sh in yz xyz z; do
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline --self-halo $sh > $OUT/bench_selfhalo_$sh.json 2> $OUT/bench_selfhalo_$sh.err
python -c "
import json;d=json.load(open('$OUT/bench_selfhalo_$sh.json'));print('selfhalo $sh', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('launch_group_ms'))"
done
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline > $OUT/bench_512.json 2> $OUT/bench_512.err; python -c "
import json;d=json.load(open('$OUT/bench_512.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launch_group_ms'])"
