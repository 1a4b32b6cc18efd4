#!/usr/bin/env python3
"""multi_rank_path leg of bench.py alone (two coupled 512^3 blocks on one device, every pipeline, in-process transport + the two parked ipc processes)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench_extras as bench
args = bench.parse_args(["--n", sys.argv[1] if len(sys.argv) > 1 else "512"])
bench.start_ipc_helpers(args)
try:
    from __graft_entry__ import load_package
    jr = load_package()
    out = bench.cfg_multi_rank_path(jr, n=args.n)
finally:
    bench.stop_ipc_helpers()
print(json.dumps(out))
