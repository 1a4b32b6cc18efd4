#!/bin/bash
mkdir -p gpurun_out/r04t
timeout 900 python -m pytest tests/test_gpu_thermal3d.py tests/test_gpu_thermal_multiphase.py tests/test_gpu_small_grid_graphs.py -q -x -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
for r in 1 2 3; do for f in 0 1; do echo "thermal_fused_batch=$f"; timeout 300 python scripts/bench3d_extra.py 0 256 thermal_fused_batch=$f 2>&1 | grep it_per_s | cut -c1-150; done; done | tee gpurun_out/r04t/fused_batch_ab.txt
for n in 64 128; do for f in 0 1; do echo "n=$n thermal_fused_batch=$f"; timeout 300 python scripts/bench3d_extra.py 0 $n thermal_fused_batch=$f 2>&1 | grep it_per_s | cut -c1-150; done; done | tee -a gpurun_out/r04t/fused_batch_ab.txt
