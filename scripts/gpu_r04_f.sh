#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_sizes.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for r in 1 2; do
for vf in 1 0; do
  timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-general-kernel --option visc_fold=$vf > $OUT/bench512_vf${vf}_$r.json 2> $OUT/bench512_vf${vf}_$r.err
  timeout 600 python3 bench.py --n 256 --steps 400 --warmup 20 --no-extras --no-cpu-baseline --no-general-kernel --option visc_fold=$vf > $OUT/bench256_vf${vf}_$r.json 2> $OUT/bench256_vf${vf}_$r.err
done
done
python3 - <<PY
import json
for n in (512, 256):
    for r in (1, 2):
        for vf in (1, 0):
            try:
                d = json.load(open("$OUT/bench%d_vf%d_%d.json" % (n, vf, r)))
                print(n, "visc_fold", vf, round(d["value"], 1), "it/s  kernel", round(d["roofline"]["avg_launch_ms"], 4), "ms  frac", round(d["roofline"]["frac"], 4), "whole", round(d["roofline"]["whole_iteration"]["frac"], 4))
            except Exception as e:
                print(n, vf, "error", e)
PY
