#!/bin/bash
# HBM-side traffic (FETCH_SIZE and WRITE_SIZE, one --pmc pass each, no tracing) per kernel of scripts/bench2d.py
#   bash scripts/pmc_traffic_2d.sh <tag> <cells> [config names / KEY=INT ... forwarded to bench2d.py]
# <cells> = cells of the grid the configs run on (e.g. 1048576 for 1024^2): the passes are bytes / (8 B x cells).
# FETCH_SIZE is doubled on output (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "HBM"); both are in KiB units from rocprofv3.
if [ $# -lt 2 ]; then echo "usage: $0 <tag> <cells> [bench2d.py arguments ...]" >&2; exit 2; fi
TAG=$1; NV=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/scripts/bench2d.py "$@" > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/scripts/bench2d.py "$@" > $OUT/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = float($NV)
print(f"# options: $@   cells = {n:.0f}; passes = bytes / (8 B x cells)")
for k, d in sorted(res.items()):
    if "at::" in k or "rocclr" in k: continue
    fe = 2.0 * 1024.0 * sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [0])), 1)
    wr = 1024.0 * sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [0])), 1)
    print(f"{k:60s} launches {len(d.get('FETCH_SIZE', [])):4d}  fetch {fe / 1e9:7.3f} GB ({fe / 8 / n:6.1f} passes)  write {wr / 1e9:7.3f} GB ({wr / 8 / n:6.1f} passes)")
PY
