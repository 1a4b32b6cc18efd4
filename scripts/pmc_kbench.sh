#!/bin/bash
# HBM traffic counters for the kbench kernels: separate --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 hit-miss)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
N=${2:-512}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$tag -- $GRAFT_REPO_ROOT/scripts/kbench $N 1 > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
out = "$OUT"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"][:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(res.items()):
    print(k, {c: (sum(v)/len(v), len(v)) for c, v in d.items()})
PY
