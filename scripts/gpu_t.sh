#!/bin/bash
OUT=gpurun_out/${1:-t}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k taylor > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -E "^E " $OUT/pytest.log | head -12
