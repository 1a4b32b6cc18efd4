#!/usr/bin/env python3
"""2D VEP (shear band) it/s at several sizes with a library switch off / on, alternating: bench_vep2d_switch.py KEY n [n ...]"""
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
for n in [int(a) for a in sys.argv[2:]] or [64, 128, 256, 512, 1024, 2048]:
    iters = max(200, min(6000, int(6e9 / n ** 2 / 10)))
    row = {"n": n, "iters": iters, "switch": key}
    for rep in range(2):
        for v in (0, 1):
            h.set_option(key, v)
            row.setdefault(f"{v}", []).append(round(bench.cfg_shearband(jr, h, n, iters)["it_per_s"], 1))
    h.set_option(key, 1)
    print(json.dumps(row), flush=True)
