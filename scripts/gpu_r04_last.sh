#!/bin/bash
# the default bench line and the driver's own command on one box, with profiles/pmc_traffic.json of the final kernels in place
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04last}
mkdir -p $OUT
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err
python - <<PY
import json
for f in ("bench_default", "bench_20"):
    d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value %.1f" % d["value"], "kernel %.3f ms frac %.3f" % (r["avg_launch_ms"], r["frac"]), "traffic", r["traffic"], r.get("traffic_ratio"), "steady", (d.get("steady_state") or {}).get("value"),
          "| with forces %.1f | general %.1f | general zero forces %.1f" % (d["with_body_forces"]["it_per_s"], d["general_kernel"]["it_per_s"], d["general_kernel_zero_forces"]["it_per_s"]))
PY
