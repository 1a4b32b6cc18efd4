#!/bin/bash
# kernel stats + HBM-side traffic of the 2D shear band leg (1024^2): bash scripts/gpu_prof_2d.sh [tag]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof2d}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/scripts/bench2d.py shearband > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^{" $OUT/stats.log | cut -c1-200
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
grep -v "at::native\|rocclr" $f | cut -c1-200 | head -8
bash scripts/pmc_traffic_2d.sh $(basename $OUT)/pmc 1048576 shearband 2>&1 | grep -v "^$" | cut -c1-200 | tail -8
