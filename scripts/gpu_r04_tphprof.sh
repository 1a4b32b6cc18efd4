#!/bin/bash
mkdir -p gpurun_out/r04t
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04t/prof -- python3 $GRAFT_REPO_ROOT/scripts/bench3d_extra.py 0 256 phases > $GRAFT_REPO_ROOT/gpurun_out/r04t/prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r04t/prof -name "*kernel_stats.csv" | head -1)
grep it_per_s $GRAFT_REPO_ROOT/gpurun_out/r04t/prof.log | cut -c1-200
cp $f $GRAFT_REPO_ROOT/gpurun_out/r04t/tph_kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:6]:
    print(r['Name'][:100].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'])
PY
