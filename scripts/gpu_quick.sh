#!/bin/bash
# quick GPU check: the variant-equality tests + short benches
OUT=gpurun_out/${1:-q}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stokes3d.py -m gpu -x -q -k "variants or iterate_timed or solve_matches or solvi3d" > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log | cut -c1-400
timeout 600 python bench.py --steps 50 --warmup 5 --n 256 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err; cat $OUT/bench_256.json | cut -c1-200; python -c "
import json;d=json.load(open('$OUT/bench_256.json'));print(d['value'], d['roofline'])"
timeout 600 python bench.py --steps 50 --warmup 5 --n 512 --no-cpu-baseline > $OUT/bench_512.json 2> $OUT/bench_512.err; python -c "
import json;d=json.load(open('$OUT/bench_512.json'));print(d['value'], d['ms_per_step'], d['roofline'])"
tail -3 $OUT/bench_512.err
