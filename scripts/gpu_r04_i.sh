#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for pct in 15 5 30 50 15; do
  timeout 400 python3 scripts/bench_ipc_first_pct.py $pct 2>> $OUT/err.txt | tee -a $OUT/first_pct.txt
done
