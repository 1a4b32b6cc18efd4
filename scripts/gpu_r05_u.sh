#!/bin/bash
mkdir -p gpurun_out/r05u
python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_two_blocks.py -m gpu -q > gpurun_out/r05u/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r05u/tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r05u/tests.log | head -8 | cut -c1-300
python - <<'PY' 2>&1 | tail -8
import sys, statistics
sys.path.insert(0, '.')
import bench
from __graft_entry__ import load_package
jr = load_package()
for name, opts in (("general form, in-kernel faces (general_hif 3)", dict(viscous_limit=0, zero_forces=0, general_hif=3)), ("general form, early exchange (general_hif 0)", dict(viscous_limit=0, zero_forces=0, general_hif=0)),
                   ("general form, default", dict(viscous_limit=0, zero_forces=0))):
    for split in ("x", "z"):
        r = bench.cfg_multi_rank_path(jr, only=(split, "default"), handle_options=opts)["block_it_per_s"]
        print(name, "split", split, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r["default"].items()}, "one block", round(r["one_block"], 1), flush=True)
PY
