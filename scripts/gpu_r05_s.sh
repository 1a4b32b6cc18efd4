#!/bin/bash
mkdir -p gpurun_out/r05s
bash scripts/clock_probe.sh 3 call_s | grep "clock probe" | cut -c1-200
python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_two_blocks.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fullsize.py -m gpu -q > gpurun_out/r05s/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r05s/tests.log | tail -2; grep -E "^FAILED" gpurun_out/r05s/tests.log | head -5
for i in 1 2; do python scripts/ab_tile.py 512 41 1 2>&1 | tail -1; python scripts/ab_tile.py 512 41 0 2>&1 | tail -1; done
python scripts/ab_tile.py 256 200 1 2>&1 | tail -1
