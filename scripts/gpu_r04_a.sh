#!/bin/bash
# round-4 check run: GPU test suite (incl. the two-process ipc test) + the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
cp -r /tmp/jrx_ipc_* $OUT/ 2>/dev/null
timeout 1200 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.json
tail -5 $OUT/bench_default.err
