#!/bin/bash
# L2 -> fabric read traffic of the 512^3 kernel per PHYSICAL placement draw: probe_skew.py (re-rolls in one process) under rocprofv3 --pmc FETCH_SIZE, per-roll launch times beside it
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05pp; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for CF in 1:64:0:32 1:1024:0:32; do
  T=$(echo $CF | tr : _)
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$T -- python3 $GRAFT_REPO_ROOT/scripts/probe_skew.py 512 10 $CF > $OUT/$T.txt 2> $OUT/$T.err
  grep placement $OUT/$T.txt | cut -c1-250
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/$T/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fused3d" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
v = [x[1] * 2 * 1024 / 1e9 for x in rows]
print(len(v), "fused launches; GB fetched per launch, mean of each group of 18 (first group 19):")
g = [v[:19]] + [v[19 + 18 * i: 19 + 18 * (i + 1)] for i in range((len(v) - 19) // 18)]
print(" ".join(f"{sum(x) / len(x):.3f}" for x in g if x))
PY
  rm -rf $OUT/$T
done
