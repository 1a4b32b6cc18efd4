#!/bin/bash
# SQ issue / stall counters of the 3D VEP kernels at 256^3 (own --pmc passes, no tracing)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcsqvep}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/scripts/bench3d_extra.py 256 0 > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
out = "$OUT"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"].replace("(anonymous namespace)::","")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(res.items()):
    if "k_vep3" in k or "k_velocity3d" in k:
        print(k, {c: round(sum(v)/len(v)) for c, v in sorted(d.items())})
PY
