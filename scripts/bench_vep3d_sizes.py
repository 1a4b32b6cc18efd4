#!/usr/bin/env python3
"""3D VEP it/s at several sizes under the forms of the edge pass (tuning switch vep3_edges): is the default (4) the right one everywhere?"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench_extras as bench
h = _lib.default_handle(0)
for n in [int(a) for a in sys.argv[1:]] or [48, 64, 96, 128, 160]:
    iters = max(60, min(1500, int(4e9 / n ** 3)))
    row = {"n": n, "iters": iters}
    for e in (4, 3, 1, 0):
        h.set_option("vep3_edges", e)
        row[f"edges{e}"] = round(bench.cfg_shearband3d(jr, h, n, iters)["it_per_s"], 1)
    h.set_option("vep3_edges", 4)
    for peel in (0,):
        h.set_option("vep3_peel", peel)
        row["edges4_nopeel"] = round(bench.cfg_shearband3d(jr, h, n, iters)["it_per_s"], 1)
    h.set_option("vep3_peel", 1)
    print(json.dumps(row), flush=True)
