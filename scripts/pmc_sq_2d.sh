#!/bin/bash
# SQ counters of the kernels of scripts/bench2d.py (own --pmc pass, no tracing)
#   bash scripts/pmc_sq_2d.sh <tag> [bench2d.py arguments ...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcx}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/scripts/bench2d.py ${@:2} > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].replace("(anonymous namespace)::","")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(res.items()):
    if "at::" in k or "rocclr" in k: continue
    g = {c: sum(v)/len(v) for c, v in d.items()}
    wc = g.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:40s} waves {g.get('SQ_WAVES',0):9.0f} valu/wave {g.get('SQ_INSTS_VALU',0)/max(g.get('SQ_WAVES',1),1):7.0f} vmem_rd/wave {g.get('SQ_INSTS_VMEM_RD',0)/max(g.get('SQ_WAVES',1),1):6.0f}  wait_any {g.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {g.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {g.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} valu_active {g.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f}")
PY
