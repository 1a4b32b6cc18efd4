#!/bin/bash
mkdir -p gpurun_out/r04f2
timeout 900 python -m pytest tests/test_gpu_stokes2d_thermal.py tests/test_gpu_thermal_multiphase.py tests/test_gpu_nonuniform.py -q -x -m gpu > gpurun_out/r04f2/pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r04f2/pytest.txt | tail -5
timeout 1200 python scripts/bench_solcx_fused.py 2>&1 | grep '"n"' | tee gpurun_out/r04f2/solcx_sizes.txt
