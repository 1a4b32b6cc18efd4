#!/bin/bash
OUT=gpurun_out/${1:-t}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_halo.py tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 600 python bench.py --steps 100 --warmup 10 --n 256 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err; python -c "
import json;d=json.load(open('$OUT/bench_256.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launch_group_ms'])"
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline > $OUT/bench_512.json 2> $OUT/bench_512.err; python -c "
import json;d=json.load(open('$OUT/bench_512.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launch_group_ms'])"
