#!/bin/bash
# rocprofv3 kernel stats of scripts/bench3d_extra.py:  bash scripts/prof_extra.sh <tag> [n_vep] [n_thermal]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-profx}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/scripts/bench3d_extra.py ${2:-256} ${3:-256} > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native\|rocclr" $f | cut -c1-200 | head -24
