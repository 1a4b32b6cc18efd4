#!/bin/bash
OUT=gpurun_out/${1:-th}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_thermal3d.py tests/test_gpu_halo.py tests/test_gpu_stokes2d_thermal.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 600 python scripts/bench3d_extra.py 256 256 2>/dev/null | tail -2 | tee $OUT/bench3d_extra.txt | cut -c1-200
timeout 600 python scripts/bench3d_extra.py 0 128 2>/dev/null | tail -1 | tee -a $OUT/bench3d_extra.txt | cut -c1-200
timeout 600 python scripts/bench3d_extra.py 0 384 2>/dev/null | tail -1 | tee -a $OUT/bench3d_extra.txt | cut -c1-200
