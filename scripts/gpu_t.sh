#!/bin/bash
OUT=gpurun_out/${1:-t}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_vep2d.py tests/test_gpu_halo.py tests/test_gpu_stokes2d_thermal.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -E "^E " $OUT/pytest.log | head -8
timeout 900 python scripts/bench2d.py 2>/dev/null | tail -6 | tee $OUT/bench2d.txt | cut -c1-220
