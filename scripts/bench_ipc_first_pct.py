#!/usr/bin/env python3
"""two 512^3 blocks as two processes (ipc transport) for one value of the tuning switch fused_first_pct:  python3 scripts/bench_ipc_first_pct.py <pct> [n]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
pct = sys.argv[1]
args = bench.parse_args(["--n", sys.argv[2] if len(sys.argv) > 2 else "512", "--option", f"fused_first_pct={pct}"])
bench.start_ipc_helpers(args)
try:
    out = bench.run_ipc_helpers()
finally:
    bench.stop_ipc_helpers()
for s in ("split_x", "split_z"):
    if s in out:
        print("first_pct", pct, s, "uncoupled", round(out[s]["two_uncoupled_blocks_block_it_per_s"], 1),
              {m: (round(out[s][m]["block_it_per_s"], 1), round(out[s][m].get("overhead_pct", 0), 2)) for m in ("default", "inkernel", "early", "serial")}, flush=True)
print(out.get("error"))
