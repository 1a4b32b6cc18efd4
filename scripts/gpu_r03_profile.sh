#!/bin/bash
# Round-3 evidence run (one gpurun call): the default bench line, rocprofv3 kernel stats of the HEADLINE leg alone (one size in the k_fused3d row),
# and the HBM-side traffic of its dominant kernel from separate --pmc passes.   bash scripts/gpu_r03_profile.sh [tag]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-steady-state > $OUT/headline_profiled_run.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 20 --warmup 2 > $OUT/$C.json 2> $OUT/$C.err
done
cd $GRAFT_REPO_ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/bench_headline_kernel_stats.csv
grep -v "at::native\|rocclr" $f | cut -c1-220 | head -8
python3 - <<PY > $OUT/pmc_bench_traffic.txt
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/FETCH_SIZE/*/*counter_collection.csv") + glob.glob("$OUT/WRITE_SIZE/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 512.0 ** 3
print("# python3 bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 20 --warmup 2 under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each); FETCH_SIZE doubled")
print("# (gfx950: 128-B read requests tallied at 64 B, MI355X_MICROARCH.md 'HBM'); rocprofv3 reports KiB; passes = bytes / (8 B x 512^3)")
print("# k_fused3d<..., TAG, VISC>: last template argument true = viscous-limit form (the headline, dt = Inf), false = general form (the general_kernel leg)")
for k, d in sorted(res.items()):
    if "at::" in k or "rocclr" in k: continue
    fv, wv = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    fe = 2.0 * 1024.0 * sum(fv) / max(len(fv), 1)
    wr = 1024.0 * sum(wv) / max(len(wv), 1)
    print(f"{k:100s} launches {len(fv):4d}  fetch {fe / 1e9:8.3f} GB ({fe / 8 / n:5.1f} passes)  write {wr / 1e9:8.3f} GB ({wr / 8 / n:5.1f} passes)  total {(fe + wr) / 1e9:8.3f} GB")
PY
grep -v "^#" $OUT/pmc_bench_traffic.txt | head -8
