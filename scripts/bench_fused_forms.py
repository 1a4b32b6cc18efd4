#!/usr/bin/env python3
"""Kernel time of every form of k_fused3d on one box, one process, same allocations, alternating: general / viscous-limit form x body forces loaded / ρg_x, ρg_y not loaded
(gravity along z: ρg_z = 1 here) / none loaded (SolVi3D's zeros).  SolVi3D n^3, dt = Inf; the general form is forced with option viscous_limit = 0.
    python3 scripts/bench_fused_forms.py [n] [steps]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
jr = load_package()
import torch  # noqa: E402
from justrelax_jl_amd import _lib, stokes  # noqa: E402
import justrelax_jl_amd.grid as grid  # noqa: E402
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device  # noqa: E402
import bench_extras as bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 41
h = _lib.default_handle(0)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
run(5)
cells = float(n) ** 3
forms = [("general, forces loaded", 0, 0, 0.0), ("general, gravity along z", 0, 1, 1.0), ("general, no forces", 0, 1, 0.0),
         ("viscous limit, forces loaded", 1, 0, 0.0), ("viscous limit, gravity along z", 1, 1, 1.0), ("viscous limit, no forces", 1, 1, 0.0)]
for rnd in range(3):
    for name, visc, zf, fz in forms:
        h.set_option("viscous_limit", visc)
        h.set_option("zero_forces", zf)
        ρg[2].fill_(fz)
        torch.cuda.synchronize()
        f0 = bench.counters(h)
        r = run(steps)
        pr = bench.pricing(h, dt, bench.nof_ran(h, f0))
        ms = r[4]
        print(f"n {n} round {rnd}  {name:32s} kernel {ms:7.4f} ms  = {1e3 / ms:7.1f} launches/s  priced {pr['alg']:.0f} B/cell  frac {pr['alg'] * cells / (ms * 1e-3) / 1e9 / 8000.0:.3f}  ({pr['form']})", flush=True)
ρg[2].fill_(0.0)
h.set_option("viscous_limit", 1); h.set_option("zero_forces", 1)
