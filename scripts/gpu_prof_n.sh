#!/bin/bash
# rocprofv3 kernel stats of the headline leg at another block size: bash scripts/gpu_prof_n.sh <n> [steps]
N=${1:-256}; K=${2:-400}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_n$N
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --n $N --steps $K --warmup 20 --no-extras --no-cpu-baseline --no-general-kernel > $OUT/run.json 2> $OUT/err.txt
cd $GRAFT_REPO_ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
cut -c1-200 $OUT/kernel_stats.csv | head -8
python3 -c "
import json; d=json.load(open('$OUT/run.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launch_group_ms'])"
