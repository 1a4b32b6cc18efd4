#!/bin/bash
OUT=gpurun_out/${1:-vep}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_vep3d.py tests/test_gpu_halo.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -3
grep -E "^E " $OUT/pytest.log | head -8
for m in 1 2; do
timeout 600 python scripts/bench3d_extra.py 256 0 2>/dev/null | tail -1 | cut -c1-200
done | tee $OUT/vep.log
