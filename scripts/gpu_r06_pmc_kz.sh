#!/bin/bash
# where does the headline kernel's excess fetch come from?  FETCH_SIZE per launch for chunk depths 8 / 12 (prologue plane: 1/8 against 1/12) and the 64 x 4 tile (y halo 4/3 against 8/7), at 512^3 and 256^3
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $OUT
for n in 512 256; do for v in base kz8 tile0; do
  case $v in base) o="";; kz8) o="--option fused_kz=8";; tile0) o="--option fused_tile=0";; esac
  [ $n = 256 ] && [ $v = kz8 ] && o="--option fused_kz=12"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${v}_$n -- python3 $GRAFT_REPO_ROOT/bench.py --n $n --no-cpu-baseline --no-steady-state --no-state-check --no-general-kernel --steps 20 --warmup 2 $o --details $OUT/${v}_$n.details.json > $OUT/${v}_$n.json 2> $OUT/${v}_$n.err
  python3 - $OUT/${v}_$n "$n $v $o" $OUT/${v}_$n.json <<'P'
import csv,glob,sys,collections,json
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fused3d" in r["Kernel_Name"]: d[r["Kernel_Name"].split("k_fused3d")[1][:40]].append(float(r["Counter_Value"]))
ms=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["roofline"]["avg_launch_ms"]
n=int(sys.argv[2].split()[0])
for k,v in d.items(): print(f"{sys.argv[2]:32s} {k}: launches {len(v)} fetch {2*1024*sum(v)/len(v)/1e9:.3f} GB per launch = {2*1024*sum(v)/len(v)/8/n**3:.2f} passes; {ms:.3f} ms (under the profiler)")
P
done; done | tee $OUT/summary.txt
