#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04x2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for cfg in "130 40 35 1 0" "130 40 35 1 1" "130 40 35 1 2" "130 40 35 1 3" "70 13 24 1 0" "61 10 16 1 3" "200 64 40 1 3"; do
  echo "== $cfg"; timeout 120 scripts/kbench_x2 $cfg 2>&1 | tail -14
done > $OUT/x2_small.log 2>&1
cat $OUT/x2_small.log | head -120
timeout 600 scripts/kbench_x2 512 512 512 10 0 > $OUT/x2_512.log 2>&1
cat $OUT/x2_512.log
timeout 300 scripts/kbench_x2 256 256 256 30 0 > $OUT/x2_256.log 2>&1
grep -v mismatch $OUT/x2_256.log
