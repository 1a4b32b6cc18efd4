#!/bin/bash
mkdir -p gpurun_out/r04n
timeout 900 python -m pytest tests/test_gpu_vep2d.py tests/test_gpu_vep_extras.py tests/test_gpu_small_grid_graphs.py tests/test_gpu_nonuniform.py -q -x -m gpu > gpurun_out/r04n/pytest_vep2d.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r04n/pytest_vep2d.txt | tail -4
timeout 900 python scripts/bench_vep2d_switch.py fused2d_batch 64 128 256 512 1024 2048 2>&1 | grep '"n"' | tee gpurun_out/r04n/batch_2d.txt
