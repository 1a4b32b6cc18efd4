#!/bin/bash
# is the first process of a fresh box slower, and does pre-allocating one large slab (bench.py --slab-gb) change it?   bash scripts/gpu_r04_slab.sh tag "60 0 60 0 0"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04slab}
mkdir -p $OUT
r=0
for sl in $2; do
  r=$((r+1))
  timeout 600 python bench.py --slab-gb $sl --no-extras --no-cpu-baseline --no-general-kernel --no-steady-state --steps 40 --warmup 5 > $OUT/b_$r.json 2> $OUT/b_$r.err
  python - <<PY
import json
d = json.loads(open("$OUT/b_$r.json").read().strip().splitlines()[-1])
print("process $r  slab $sl GB: value %.1f it/s  kernel %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]))
PY
done
