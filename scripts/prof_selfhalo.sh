#!/bin/bash
# rocprofv3 kernel stats of the N > 1 code path on one device (bench.py --self-halo [dims])
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-profsh}
SH=${2:-xyz}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --n 512 --no-cpu-baseline --self-halo $SH > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native\|rocclr" $f | cut -c1-230 | head -16
