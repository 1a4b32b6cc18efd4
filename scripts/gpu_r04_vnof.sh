#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04vnof}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_vep3d.py tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_sizes.py tests/test_gpu_vep_extras.py -m gpu -q > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -2; grep -E "^FAILED|AssertionError: " $OUT/pytest.txt | head
timeout 600 python3 scripts/bench_vep3d_switch.py zero_forces 256 > $OUT/vep.txt 2>$OUT/vep.err; cat $OUT/vep.txt
timeout 600 python3 scripts/bench_end_flips.py 512 20 > $OUT/b20.txt 2>> $OUT/vep.err; cat $OUT/b20.txt
