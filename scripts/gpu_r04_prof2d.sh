#!/bin/bash
mkdir -p gpurun_out/r04p2
cd /tmp && export TMPDIR=/tmp
for cfg in "shearband 1024 600" "solcx 1536 800" "shearband 256 3000"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04p2/$tag -- python3 $GRAFT_REPO_ROOT/scripts/bench2d_one.py $cfg > $GRAFT_REPO_ROOT/gpurun_out/r04p2/$tag.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r04p2/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; grep it_per_s $GRAFT_REPO_ROOT/gpurun_out/r04p2/$tag.log | cut -c1-160
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r04p2/$tag.csv && head -7 "$f" | cut -c1-110,200-300
done
