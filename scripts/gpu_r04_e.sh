#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vep3d.py tests/test_gpu_fullsize.py tests/test_gpu_vep_extras.py tests/test_gpu_creep.py tests/test_gpu_two_blocks.py tests/test_gpu_small_grid_graphs.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
timeout 600 python3 scripts/bench_vep_fork.py 256 3 > $OUT/vep_fork_256.txt 2>&1
grep "n=" $OUT/vep_fork_256.txt
timeout 300 python3 scripts/bench_vep_fork.py 160 2 > $OUT/vep_fork_160.txt 2>&1
grep "n=" $OUT/vep_fork_160.txt
