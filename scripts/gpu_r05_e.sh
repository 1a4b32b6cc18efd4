#!/bin/bash
mkdir -p gpurun_out/r05e
python -m pytest tests/test_gpu_field_alloc.py tests/test_gpu_stokes3d.py tests/test_gpu_two_blocks.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fullsize.py tests/test_gpu_ipc_two_processes.py tests/test_gpu_halo.py -m gpu -q > gpurun_out/r05e/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r05e/tests.log | tail -3; grep -E "^FAILED" gpurun_out/r05e/tests.log | head
for i in 1 2 3 4; do python scripts/ab_tile.py 512 41 2>&1 | tail -1; done
for i in 1 2 3; do python scripts/ab_tile.py 256 200 2>&1 | tail -1; done
python scripts/ab_tile.py 384 60 2>&1 | tail -1
