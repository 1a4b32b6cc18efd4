#!/bin/bash
# the default bench line three times in a row on one box (three processes): what the placement of a process's allocations does to the same binary
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04spread}
mkdir -p $OUT
for r in 1 2 3; do
  timeout 1200 python bench.py > $OUT/bench_default_$r.json 2> $OUT/bench_default_$r.err
done
python - <<PY
import json
for r in (1, 2, 3):
    d = json.loads(open("$OUT/bench_default_%d.json" % r).read().strip().splitlines()[-1])
    ro = d["roofline"]
    print(r, "value %.1f" % d["value"], "kernel %.3f ms frac %.3f" % (ro["avg_launch_ms"], ro["frac"]), "traffic ratio", ro.get("traffic_ratio"),
          "| with forces %.1f | general %.1f | general zero forces %.1f | solve %.1f | 256^3 %.0f | vep3d %.0f" % (d["with_body_forces"]["it_per_s"], d["general_kernel"]["it_per_s"],
          d["general_kernel_zero_forces"]["it_per_s"], d["solve_path"]["it_per_s"], d["other_configs"]["solvi3d_256"]["it_per_s"], d["other_configs"]["shearband3d_256"]["it_per_s"]))
PY
