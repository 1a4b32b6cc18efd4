#!/usr/bin/env python3
"""3D VEP 256^3 it/s under single tuning switches (one process, baseline re-measured between them)"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench
h = _lib.default_handle(0)
def run():
    return round(bench.cfg_shearband3d(jr, h)["it_per_s"], 1)
print(json.dumps({"baseline": [run(), run()]}), flush=True)
for key, vals in (("vep3_nt", (1,)), ("vep3_prekz", (4, 16, 32)), ("vep3_xcd", (0,)), ("vep3_map", (0,)), ("vep3_peel", (0,)), ("vep3_peel_fork", (1,)), ("vep3_edges", (3, 1))):
    d0 = h.get_option(key)
    for v in vals:
        h.set_option(key, v)
        print(json.dumps({key: v, "it_per_s": [run(), run()], "default": d0}), flush=True)
    h.set_option(key, d0)
    print(json.dumps({"baseline": [run()]}), flush=True)
