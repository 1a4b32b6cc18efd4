#!/bin/bash
mkdir -p gpurun_out/r04t
timeout 900 python -m pytest tests/test_gpu_thermal_multiphase.py tests/test_gpu_thermal3d.py -q -x -m gpu 2>&1 | tail -5
for r in 1 2 3; do for f in 0 1; do echo "thermal_np_const=$f"; timeout 300 python scripts/bench3d_extra.py 0 256 phases thermal_np_const=$f 2>&1 | grep it_per_s | cut -c1-170; done; done | tee gpurun_out/r04t/ab.txt
