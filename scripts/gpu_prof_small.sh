#!/bin/bash
# rocprofv3 kernel stats of the 3D loops on a small grid: bash scripts/gpu_prof_small.sh <n>
N=${1:-64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_small$N
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/scripts/bench_small3d.py $N > $OUT/run.txt 2> $OUT/err.txt
cd $GRAFT_REPO_ROOT
cat $OUT/run.txt
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
grep -v "at::native" $OUT/kernel_stats.csv | cut -c1-150 | awk -F'","' '{print $1 " | calls " $2 " | avg ns " $4}' | head -24
