#!/usr/bin/env python3
"""In-process A/B of the tuning switch end_flips (jrx_stokes3d_iterate_timed: out-of-place end sweeps instead of an un-fused first iteration when the number of fused steps is odd):
wall time of 20-step batches of SolVi3D n^3, alternating, same allocations.   python3 scripts/bench_end_flips.py [n] [steps]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
jr = load_package()
import torch  # noqa: E402
from justrelax_jl_amd import _lib, stokes  # noqa: E402
import justrelax_jl_amd.grid as grid  # noqa: E402
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h = _lib.default_handle(0)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
run(5)
torch.cuda.synchronize()
res = {0: [], 1: []}
for rnd in range(6):
    for fl in (1, 0):
        h.set_option("end_flips", fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        res[fl].append((time.perf_counter() - t0) * 1e3)
h.set_option("end_flips", 1)
for fl in (1, 0):
    v = sorted(res[fl])
    print(f"n {n} steps {steps} end_flips {fl}: batch ms " + " ".join(f"{x:.2f}" for x in res[fl]) + f"   median {v[len(v) // 2]:.2f} ms = {steps / v[len(v) // 2] * 1e3:.1f} it/s")
