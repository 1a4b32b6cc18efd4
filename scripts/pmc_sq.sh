#!/bin/bash
# SQ issue/stall counters for the kbench kernels (own --pmc passes, no tracing)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcsq}
N=${2:-512}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$tag -- $GRAFT_REPO_ROOT/scripts/kbench $N 1 > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
out = "$OUT"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"][:130]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(res.items()):
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
