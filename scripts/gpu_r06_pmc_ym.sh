#!/bin/bash
# L2 <- fabric read traffic (FETCH_SIZE) of the headline kernel: one tile per block against the y march (fused_ym = 2, 4) and the 64 x 16 tile
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $OUT
for v in base ym2 ym4 t16; do
  case $v in base) o="";; ym2) o="--option fused_ym=2";; ym4) o="--option fused_ym=4";; t16) o="--option fused_tile=4";; esac
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-steady-state --no-state-check --no-general-kernel --steps 20 --warmup 2 $o --details $OUT/$v.details.json > $OUT/$v.json 2> $OUT/$v.err
  python3 - $OUT/$v $v <<'P'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fused3d" in r["Kernel_Name"]: d[r["Kernel_Name"][-60:]].append(float(r["Counter_Value"]))
for k,v in d.items(): print(f"{sys.argv[2]:5s} {k}: launches {len(v)} fetch {2*1024*sum(v)/len(v)/1e9:.3f} GB per launch")
P
done | tee $OUT/summary.txt
