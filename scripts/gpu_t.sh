#!/bin/bash
OUT=gpurun_out/${1:-t}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_halo.py tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -E "^E " $OUT/pytest.log | head -8
for n in 256 384 512; do
timeout 600 python bench.py --steps 100 --warmup 10 --n $n --no-cpu-baseline > $OUT/bench_$n.json 2> $OUT/bench_$n.err; python -c "
import json;d=json.load(open('$OUT/bench_$n.json'));print($n, round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['launch_group_ms'],3), round(d['roofline']['frac'],3))"
done
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline --self-halo xyz > $OUT/bench_sh.json 2> $OUT/bench_sh.err; python -c "
import json;d=json.load(open('$OUT/bench_sh.json'));print('selfhalo xyz', round(d['value'],2), round(d['ms_per_step'],3))"
