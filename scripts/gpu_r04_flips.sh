#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04flip}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_sizes.py -m gpu -x -q -k "iterate_timed or fullsize or solvi" > $OUT/pytest.txt 2>&1
grep -E "passed|failed|error|Error|assert" $OUT/pytest.txt | tail -8
for fl in 1 0 1 0; do
  timeout 600 python bench.py --option end_flips=$fl --no-extras --no-cpu-baseline --no-general-kernel --no-steady-state --steps 20 --warmup 5 > $OUT/bench_$fl.json 2> $OUT/bench_$fl.err
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$fl.json").read().strip().splitlines()[-1])
print("end_flips $fl: value %.1f it/s  ms/step %.3f  kernel %.3f ms" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]), d.get("kernel_launch_counters"))
PY
done
