#!/usr/bin/env python3
"""it/s of the 3D loops on small grids with and without graph replay of the unobserved iterations (option loop_graphs).
usage: bench_small3d.py [n ...]   (default 16 32 64 96 128)"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
from justrelax_jl_amd.arrays import from_numpy
from justrelax_jl_amd.miniapps.common import upload_stokes

dev = torch.device("cuda", 0)
h = _lib.default_handle(0)


def timed(fn, warm, iters):
    fn(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(iters)
    torch.cuda.synchronize()
    return iters / (time.perf_counter() - t0)


def stokes3d(n, iters):
    s = jr.miniapps.solvi3d(n, iterMax=iters - 1, nout=10 ** 9)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    return lambda k: jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=dict(iterMax=k - 1, nout=10 ** 9, verbose=False))


def vep3d(n, iters):
    s = jr.miniapps.shearband3d(n, iterMax=iters - 1, nout=10 ** 9)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, t in dict(Vx=st.V.Vx, Vy=st.V.Vy, Vz=st.V.Vz, eta=st.viscosity.η).items():
        t.copy_(from_numpy(s.arrays[k], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    for k, name in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
        getattr(pr, name).copy_(from_numpy(s.arrays[k], dev))
    ρg = tuple(jr.fzeros(s.ni, dev) for _ in range(3))
    return lambda k: jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=dict(iterMax=k - 1, nout=10 ** 9, verbose=False))


def thermal3d(n, iters):
    s = jr.miniapps.diffusion3d(n, iterMax=iters, nout=10 ** 9)
    th = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    th.T.copy_(from_numpy(s.arrays["T"], dev)); th.H.fill_(1e-6)
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-300)
    return lambda k: jr.heatdiffusion_PT_(th, pt, s.flow_bcs, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=k, nout=10 ** 9, verbose=False))


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 96, 128]
    for n in sizes:
        iters = 4000 if n <= 64 else 1500
        for name, mk in (("stokes3d", stokes3d), ("vep3d", vep3d), ("thermal3d", thermal3d)):
            res = {}
            for g in (0, 1, 0, 1):
                h.set_option("loop_graphs", g)
                run = mk(n, iters)
                res.setdefault(g, []).append(timed(run, 200, iters))
            print(json.dumps(dict(path=name, n=n, plain_it_per_s=[round(v) for v in res[0]], graphs_it_per_s=[round(v) for v in res[1]],
                                  gain_pct=round((max(res[1]) / max(res[0]) - 1) * 100, 1))), flush=True)
    h.set_option("loop_graphs", 1)
