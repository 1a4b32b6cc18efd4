#!/bin/bash
# rocprofv3 kernel stats of the bench command:  bash scripts/prof_bench.sh <tag> <n> [steps]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
N=${2:-512}; K=${3:-20}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps $K --warmup 2 --n $N --no-cpu-baseline > $OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
grep -v "at::native\|rocclr" $f | cut -c1-260 | head -12
