#!/usr/bin/env python3
"""3D heat diffusion 256^3 it/s under the tile-shape / XCD-band switches of k_thermal3d_fused (one process)"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench
h = _lib.default_handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def run():
    return round(bench.cfg_thermal3d(jr, h, n)["it_per_s"], 1)
print(json.dumps({"n": n, "default": [run(), run()]}), flush=True)
for cfg, xg in ((10204, 1), (10204, 2), (10204, 4), (10204, 8), (10202, 1), (10202, 2), (10208, 1), (10208, 8), (10104, 1), (10104, 8), (10004 + 100 * 1 - 100 + 0, 8)):
    try:
        h.set_option("thermal_cfg", cfg); h.set_option("thermal_xg", xg)
        print(json.dumps({"thermal_cfg": cfg, "thermal_xg": xg, "it_per_s": [run(), run()]}), flush=True)
    except Exception as e:
        print(json.dumps({"thermal_cfg": cfg, "thermal_xg": xg, "error": str(e)[:80]}), flush=True)
h.set_option("thermal_cfg", 0); h.set_option("thermal_xg", 8)
print(json.dumps({"default_again": [run()]}), flush=True)
