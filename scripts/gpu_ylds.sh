#!/bin/bash
OUT=gpurun_out/${1:-fc}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_halo.py tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
for sh in xyz yz; do
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline --self-halo $sh > $OUT/bench_selfhalo_$sh.json 2> $OUT/bench_selfhalo_$sh.err
python -c "
import json;d=json.load(open('$OUT/bench_selfhalo_$sh.json'));print('selfhalo $sh', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('launch_group_ms'))"
done
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline > $OUT/bench_512.json 2> $OUT/bench_512.err; python -c "
import json;d=json.load(open('$OUT/bench_512.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launch_group_ms'])"
