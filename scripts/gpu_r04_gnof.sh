#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04gnof}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_two_blocks.py -m gpu -x -q -k "body_forces or viscous_limit or iterate_timed or two_blocks_equal" > $OUT/pytest.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest.txt | tail -3
timeout 900 python3 scripts/bench_fused_forms.py 512 41 > $OUT/forms.txt 2> $OUT/forms.err
timeout 600 python3 scripts/bench_fused_forms.py 256 201 >> $OUT/forms.txt 2>> $OUT/forms.err
cat $OUT/forms.txt; tail -2 $OUT/forms.err
