#!/bin/bash
# kernel stats of the coupled two-block leg (in-process transport), x split and z split: what runs beside k_fused3d and for how long
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05tb; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for S in x z; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$S -- python3 $GRAFT_REPO_ROOT/scripts/bench_multi_rank.py 512 20 $S default > $OUT/$S.json 2> $OUT/$S.err
  f=$(find $OUT/$S -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_$S.csv
  echo "== split $S"; grep -v "at::native\|rocclr" $f | cut -d, -f1-4 --output-delimiter=' | ' | sed 's/(anonymous namespace):://g' | cut -c1-170 | head -16
  grep -o '"overhead_pct": [0-9.-]*' $OUT/$S.json | head -2
  rm -rf $OUT/$S
done
