#!/bin/bash
# the forms of the one-launch viscous-limit kernel without body-force loads (k_fused3d, NOF): parity tests, then the headline leg with its `with_body_forces` and `general_kernel` twins
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04nof}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_two_blocks.py -m gpu -x -q -k "body_forces or viscous_limit" > $OUT/pytest.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest.txt | tail -3
timeout 600 python bench.py --no-extras --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["value"], d["config"]["kernel_form"], r["bytes_per_cell"], "kernel ms", r.get("avg_launch_ms"), "frac", r["frac"], d.get("kernel_launch_counters"))
for k in ("with_body_forces", "general_kernel"):
    g = d.get(k, {})
    print(k, g.get("it_per_s"), (g.get("roofline") or {}).get("avg_launch_ms"), (g.get("roofline") or {}).get("frac"), g.get("error"))
print("steady", d.get("steady_state"))
PY
