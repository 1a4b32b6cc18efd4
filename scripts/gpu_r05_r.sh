#!/bin/bash
mkdir -p gpurun_out/r05r
bash scripts/clock_probe.sh 3 call_r | grep "clock probe" | cut -c1-260
python -m pytest tests/test_gpu_two_blocks.py tests/test_gpu_vep3d.py tests/test_gpu_halo.py -m gpu -q -k "vep3d or vep or halo" > gpurun_out/r05r/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r05r/tests.log | tail -2; grep -E "^FAILED" gpurun_out/r05r/tests.log | head -5
python - <<'PY' 2>&1 | tail -8
import json, sys
sys.path.insert(0, '.')
import bench
from __graft_entry__ import load_package
jr = load_package()
r = bench.cfg_multi_rank_path(jr, only=("vep", "z"))
b = r["block_it_per_s"]
import statistics
for k in ("serial", "hidden", "hidden_eta_tau_only"):
    pairs = b[k]
    ov = sorted((u / c - 1) * 100 for c, u in pairs)
    print(k, "coupled", [round(c, 1) for c, _ in pairs], "uncoupled", [round(u, 1) for _, u in pairs], "overhead % median", round(statistics.median(ov), 2))
print("one block", round(b["one_block"], 1))
PY
