#!/usr/bin/env python3
"""3D VEP it/s at several sizes with a library switch off / on, alternating: bench_vep3d_switch.py KEY n [n ...]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench_extras as bench
h = _lib.default_handle(0)
key = sys.argv[1]
for n in [int(a) for a in sys.argv[2:]] or [16, 32, 48, 64, 96, 128]:
    iters = max(60, min(3000, int(4e9 / n ** 3)))
    row = {"n": n, "iters": iters, "switch": key}
    for rep in range(2):
        for v in (0, 1):
            h.set_option(key, v)
            row.setdefault(f"{v}", []).append(round(bench.cfg_shearband3d(jr, h, n, iters)["it_per_s"], 1))
    h.set_option(key, 1)
    print(json.dumps(row), flush=True)
