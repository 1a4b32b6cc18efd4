#!/bin/bash
# Round-3 evidence for the 3D VEP path at 256^3: rocprofv3 kernel stats and one FETCH_SIZE / WRITE_SIZE pass per kernel.
#   bash scripts/gpu_r03_vep_profile.sh [tag]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03vep}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/scripts/bench3d_extra.py 256 0 > $OUT/stats.log 2>&1
cd $GRAFT_REPO_ROOT
grep it_per_s $OUT/stats.log | cut -c1-200
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/vep3d_256_kernel_stats.csv
grep -v "at::native\|rocclr" $f | cut -c1-200 | head -9
bash scripts/pmc_traffic_extra.sh $(basename $OUT)/pmc 256 0 > $OUT/vep3d_256_pmc_traffic.txt 2>&1
grep "k_vep3\|k_velocity3d_zb<false\|options" $OUT/vep3d_256_pmc_traffic.txt
