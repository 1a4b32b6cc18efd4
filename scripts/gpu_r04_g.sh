#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_two_blocks.py tests/test_gpu_ipc_two_processes.py tests/test_gpu_halo.py tests/test_gpu_stokes3d.py -m gpu -q -x > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt | cut -c1-400
cp -r /tmp/jrx_ipc_* $OUT/ 2>/dev/null
