#!/bin/bash
# HBM-side traffic of the two forms of k_fused3d (option viscous_limit 1 / 0) on SolVi3D 512^3, separate --pmc passes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03_visc}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/scripts/bench_viscous_limit.py 512 > $OUT/$C.log 2> $OUT/$C.err
done
cd $GRAFT_REPO_ROOT
python3 - <<PY > $OUT/pmc_traffic.txt
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/FETCH_SIZE/*/*counter_collection.csv") + glob.glob("$OUT/WRITE_SIZE/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 512.0 ** 3
print("# python3 scripts/bench_viscous_limit.py 512 under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each); FETCH_SIZE doubled")
print("# (gfx950: 128-B read requests tallied at 64 B, MI355X_MICROARCH.md 'HBM'); rocprofv3 reports KiB; passes = bytes / (8 B x 512^3)")
for k, d in sorted(res.items()):
    if "at::" in k or "rocclr" in k: continue
    fv, wv = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    fe = 2.0 * 1024.0 * sum(fv) / max(len(fv), 1)
    wr = 1024.0 * sum(wv) / max(len(wv), 1)
    print(f"{k:100s} launches {len(fv):4d}  fetch {fe / 1e9:8.3f} GB ({fe / 8 / n:5.1f} passes)  write {wr / 1e9:8.3f} GB ({wr / 8 / n:5.1f} passes)  total {(fe + wr) / 1e9:8.3f} GB")
PY
grep -v "^#" $OUT/pmc_traffic.txt | grep fused | cut -c1-260
