#!/bin/bash
mkdir -p gpurun_out/r04t
timeout 900 python -m pytest tests/test_gpu_thermal_multiphase.py tests/test_gpu_thermal3d.py -q -x -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
for r in 1 2 3; do for f in 0 1; do echo "thermal_fused_ph=$f"; timeout 300 python scripts/bench3d_extra.py 0 256 phases thermal_fused_ph=$f 2>&1 | grep it_per_s | cut -c1-150; done; done | tee gpurun_out/r04t/fused_ph_ab.txt
