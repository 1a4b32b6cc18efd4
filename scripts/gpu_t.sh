#!/bin/bash
OUT=gpurun_out/${1:-t}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py tests/test_gpu_halo.py tests/test_gpu_bcs.py tests/test_gpu_golden.py tests/test_gpu_vep3d.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -E "^E " $OUT/pytest.log | head -8
for n in 64 96 128; do
timeout 600 python bench.py --steps 200 --warmup 10 --n $n --no-cpu-baseline > $OUT/bench_$n.json 2> $OUT/bench_$n.err
python -c "
import json;d=json.load(open('$OUT/bench_$n.json'));print('n=%4d  %9.1f it/s  %8.3f ms/it'%($n,d['value'],d['ms_per_step']))"
done
