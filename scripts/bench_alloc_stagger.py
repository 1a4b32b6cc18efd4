#!/usr/bin/env python3
"""Does the spread between processes / allocations of the 512^3 kernel come from arrays that start at the same offset of the memory system's interleaving pattern?  Six of the caller's
arrays and four of the library's scratch set are exactly 2^30 bytes at 512^3.  This script rebuilds the SolVi3D problem several times in one process with (i) torch's own placement and
(ii) every array starting k * stagger bytes into a slightly larger allocation (k = running index), and the library's scratch set staggered the same way (tuning switch
scratch_stagger), alternating, and prints the kernel time of each build.     python3 scripts/bench_alloc_stagger.py [n] [stagger bytes ...]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
jr = load_package()
import torch  # noqa: E402
from justrelax_jl_amd import _lib, stokes, arrays  # noqa: E402
import justrelax_jl_amd.grid as grid  # noqa: E402
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
staggers = [int(a) for a in sys.argv[2:]] or [0, 4352, 69888, 1118464]
h = _lib.default_handle(0)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
orig = arrays.fzeros
state = {"stg": 0, "k": 0}


def fzeros_staggered(shape, device, fill: float = 0.0):
    if not state["stg"]:
        return orig(shape, device, fill)
    shape = tuple(int(s) for s in shape)
    cnt = 1
    for s in shape:
        cnt *= s
    state["k"] = (state["k"] + 1) % 32
    off = state["k"] * state["stg"] // 8
    flat = torch.full((cnt + off,), float(fill), dtype=torch.float64, device=device)
    t = flat[off:].view(shape[::-1])
    return t.permute(*range(len(shape) - 1, -1, -1))


import justrelax_jl_amd as pkg  # noqa: E402
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, "__name__", "").startswith("justrelax_jl_amd") and getattr(mod, "fzeros", None) is orig:
        mod.fzeros = fzeros_staggered
if getattr(jr, "fzeros", None) is orig:
    jr.fzeros = fzeros_staggered

for rnd in range(4):
    for stg in staggers:
        state["stg"], state["k"] = stg, 0
        h.set_option("scratch_stagger", stg)
        st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
        jr.flow_bcs_(st, bcs, handle=h)
        ητ = jr.fzeros((n, n, n), st.P.device)
        jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
        run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
        run(5)
        r = run(41)
        print(f"n {n} round {rnd} stagger {stg:8d} B: kernel {r[4]:.4f} ms   P at 0x{st.P.data_ptr():x} txx 0x{st.τ.xx.data_ptr():x}", flush=True)
        del st, ρg, K, G, ητ, run
        torch.cuda.empty_cache()
h.set_option("scratch_stagger", 0)
