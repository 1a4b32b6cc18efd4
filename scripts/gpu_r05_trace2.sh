#!/bin/bash
# two 512^3 blocks on one device, coupled (in-process transport, default pipeline) and then uncoupled IN THE SAME PROCESS, under rocprofv3 --kernel-trace: twenty iterations from the
# middle of each timed batch -- wall time per iteration pair, device busy time, and the kernels that ran
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05tr; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for S in x z; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$S -- python3 $GRAFT_REPO_ROOT/scripts/trace_two_blocks.py $S 2 40 > $OUT/$S.txt 2> $OUT/$S.err
  cat $OUT/$S.txt
  f=$(find $OUT/t_$S -name "*kernel_trace.csv" | head -1)
  python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$f"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def nbr(r):
    n=r["Kernel_Name"]
    return n.split(",")[15].strip() if "k_fused3d" in n else None
for phase, flag, lpi in (("coupled", "true", 4), ("uncoupled", "false", 2)):
    fused=[r for r in rows if nbr(r)==flag and "64, 8, 8" in r["Kernel_Name"]]
    if phase=="uncoupled":      # the launches after the last coupled one
        last=max(int(r["Start_Timestamp"]) for r in rows if nbr(r)=="true")
        fused=[r for r in fused if int(r["Start_Timestamp"])>last]
    w0=int(fused[-30*lpi]["Start_Timestamp"]); w1=int(fused[-10*lpi]["Start_Timestamp"])
    agg=collections.defaultdict(lambda:[0,0.0]); busy=[]
    for r in rows:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        if s<w0 or s>=w1: continue
        n=r["Kernel_Name"].replace("(anonymous namespace)::","")
        n=n[:n.index("(")] if "(" in n else n
        if "k_fused3d" in n: n="k_fused3d NBR="+n.split(",")[15].strip()
        agg[n][0]+=1; agg[n][1]+=(e-s)/1e3; busy.append((s,e))
    busy.sort(); tot=0; cs,ce=busy[0]
    for s,e in busy[1:]:
        if s>ce: tot+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    tot+=ce-cs
    span=(w1-w0)/1e3
    print(f"  {phase}: {span/1e3/20:.3f} ms per iteration pair, device busy {100*tot/1e3/span:.1f} %")
    for k,(c,us) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:6]:
        print(f"    {k[:66]:66s} {c/20:5.1f} launches per pair  {us/20/1e3:7.3f} ms summed per pair  ({us/c:8.1f} us each)")
PY
  rm -rf $OUT/t_$S
done
