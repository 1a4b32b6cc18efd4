#!/bin/bash
# round 5, call A: the field pool on hardware (tests), then the placement A/B at 512^3 (fresh process per sample)
mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_field_alloc.py tests/test_gpu_stokes3d.py tests/test_gpu_thermal3d.py -m gpu -x -q > gpurun_out/r05a/tests.log 2>&1
tail -5 gpurun_out/r05a/tests.log
timeout 1500 python scripts/placement_ab.py 512 1300 > gpurun_out/r05a/placement_ab.txt 2>&1
cat gpurun_out/r05a/placement_ab.txt
