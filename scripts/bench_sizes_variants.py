#!/usr/bin/env python3
"""it/s of SolVi3D n^3 with kernel_variant 0 (auto), 2 (two sweeps), 3 (fused wherever legal): does the auto rule pick the faster path?
usage: bench_sizes_variants.py [n ...]"""
import ctypes as C
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench_extras as bench

h = _lib.default_handle(0)
import os
visc = int(os.environ.get("VISC", "1"))
h.set_option("viscous_limit", visc)
if "TILE" in os.environ:
    h.set_option("fused_tile", int(os.environ["TILE"]))
for n in [int(a) for a in sys.argv[1:]] or [64, 96, 128, 160, 192, 224, 256, 320, 384]:
    steps = max(60, min(2000, int(3e10 / n ** 3)))
    row = {"n": n, "viscous_limit": visc, "steps": steps}
    for variant in (0, 2, 3):
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
        n0 = h.get_option("stat_fused3d")
        r = bench.cfg_solvi(jr, h, n, steps, 10)
        row[f"v{variant}"] = round(r["it_per_s"], 1)
        if variant == 0:
            row["auto_fused"] = h.get_option("stat_fused3d") > n0
    print(json.dumps(row), flush=True)
h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
