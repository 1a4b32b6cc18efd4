#!/usr/bin/env python3
"""how much the nearly empty last x-tile column of k_fused3d costs: SolVi3D on (nx, n, n) for nx = exact multiples of 62 and the cubic sizes; ms per iteration
and per cell.  usage: exp_ragged_x.py n nx1 nx2 ..."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib, stokes
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
import justrelax_jl_amd.grid as grid

n = int(sys.argv[1])
h = _lib.default_handle(0)
dev = torch.device("cuda", 0)
for nx in map(int, sys.argv[2:]):
    ni = (nx, n, n)
    grid.init_global_grid(*ni)
    li = (10.0, 10.0 * n / nx, 10.0 * n / nx)
    st = jr.StokesArrays(jr.AMDGPUBackend, ni)
    geo = jr.Geometry(ni, li, origin=(0.0, 0.0, 0.0))
    pt = jr.PTStokesCoeffs(li, tuple(l / m for l, m in zip(li, ni)), CFL=1 / 3 ** 0.5)
    st.viscosity.η.fill_(1.0)
    for t in (st.V.Vx, st.V.Vy, st.V.Vz):
        t.copy_(torch.rand(tuple(t.shape), device=dev, dtype=torch.float64) * 1e-3)
    ρg = tuple(jr.fzeros(ni, dev) for _ in range(3))
    K, G, dt = jr.fzeros(ni, dev, float("inf")), jr.fzeros(ni, dev, 1.0), float("inf")
    bcs = jr.VelocityBoundaryConditions(free_slip={f: True for f in ("left", "right", "front", "back", "top", "bot")},
                                        no_slip={f: False for f in ("left", "right", "front", "back", "top", "bot")})
    jr.flow_bcs_(st, bcs, handle=h)
    ητ = jr.fzeros(ni, dev)
    jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
    run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
    run(10)
    res = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(100); torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) * 10)
    cells = nx * n * n
    print(f"nx={nx:4d} n={n}: ms/it {min(res):.4f}  kernel ms {r[4] if r[4] else 0:.4f}  ns per kcell {min(res) * 1e6 / cells * 1e3:.3f}", flush=True)
    del st, ρg, K, G, ητ
    torch.cuda.empty_cache()
