#!/bin/bash
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
for rep in 1 2; do for ov in 0 1; do for sh in xyz x; do
JRX_FUSED_OVERLAP=$ov timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline --self-halo $sh > $OUT/b_${sh}_ov${ov}_$rep.json 2> $OUT/b_${sh}_ov${ov}_$rep.err
python -c "
import json;d=json.load(open('$OUT/b_${sh}_ov${ov}_$rep.json'));print('selfhalo $sh overlap=$ov rep $rep', round(d['value'],2), round(d['ms_per_step'],3))"
done; done; done
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline > $OUT/bench_512.json 2> $OUT/bench_512.err; python -c "
import json;d=json.load(open('$OUT/bench_512.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launch_group_ms'])"
