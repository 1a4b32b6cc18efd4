#!/bin/bash
# what the driver runs at round end: smoke(), then the bench line with its own flags
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04drv}
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_20.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], d["config"]["kernel_form"], "kernel", r["avg_launch_ms"], "frac", r["frac"], "traffic", r["traffic"], r.get("traffic_ratio"))
print("steady", d["steady_state"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
print({k: (v.get("it_per_s") if isinstance(v, dict) else v) for k, v in (d.get("other_configs") or {}).items()})
PY
