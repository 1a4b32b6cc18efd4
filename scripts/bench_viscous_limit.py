#!/usr/bin/env python3
"""A/B of option viscous_limit on SolVi3D (dt = Inf): the fused kernel (kernel_variant 3) and the two z-marching sweeps (2) with and without the loads
of τ_o, P0, K, G, Q, same process.  usage: bench_viscous_limit.py [n ...]   (default 512 256)"""
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
for n in [int(a) for a in sys.argv[1:]] or [512, 256]:
    steps = 60 if n >= 512 else 300
    for variant in (3, 2):
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
        for rep in range(2):
            for visc in (1, 0):
                h.call("jrx_set_option", C.c_char_p(b"viscous_limit"), C.c_int64(visc))
                r = bench.cfg_solvi(jr, h, n, steps, 6)
                print(json.dumps({"n": n, "kernel_variant": variant, "viscous_limit": visc, "it_per_s": round(r["it_per_s"], 2), "k_fused3d_ms": round(r.get("avg_launch_ms", 0.0), 4)}), flush=True)
h.call("jrx_set_option", C.c_char_p(b"viscous_limit"), C.c_int64(1))
h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
