#!/usr/bin/env python3
"""it/s of the 2D configs of BASELINE.json on one GPU: SolCx 512^2 (config 2), shear band 1024^2 (config 5),
thermal diffusion 256^2 (config 1).  Fixed iteration counts (convergence disabled); prints one JSON line per config."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd.miniapps.common import upload_stokes
from justrelax_jl_amd.arrays import from_numpy

dev = torch.device("cuda", 0)


def timed(fn, iters):
    fn(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0), r


def solcx(n=512, iters=2000):
    s = jr.miniapps.solcx2d(n, iterMax=iters - 1, nout=10 ** 9)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    def run(k):
        return jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=dict(iterMax=k - 1, nout=10 ** 9, verbose=False))
    el, r = timed(run, iters)
    cells = n * n
    print(json.dumps(dict(config="SolCx 2D visco-elastic", n=n, iters=r.iter, it_per_s=r.iter / el, device_it_per_s=r.iter / r.time if r.time else None,
                          eff_GBps_at_240B=240.0 * cells * r.iter / el / 1e9)))


def shearband(n=1024, iters=1000):
    s = jr.miniapps.shearband2d(n, iterMax=iters - 1, nout=10 ** 9)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in dict(Vx="V.Vx", Vy="V.Vy", eta="viscosity.η").items():
        o = st
        for p in path.split("."):
            o = getattr(o, p)
        o.copy_(from_numpy(s.arrays[k], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    pr.center.copy_(from_numpy(s.arrays["phase_c"], dev)); pr.vertex.copy_(from_numpy(s.arrays["phase_v"], dev))
    ρg = (jr.fzeros(s.ni, dev), jr.fzeros(s.ni, dev))
    def run(k):
        return jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None,
                         kwargs=dict(iterMax=k - 1, nout=10 ** 9, iterMin=10 ** 9, verbose=False))
    el, r = timed(run, iters)
    cells = n * n
    print(json.dumps(dict(config="shear band 2D multiphase VEP", n=n, iters=r.iter, it_per_s=r.iter / el,
                          eff_GBps_at_700B_as_written=700.0 * cells * r.iter / el / 1e9)))


def thermal(n=256, iters=4000):
    from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
    s = jr.miniapps.diffusion2d(n, iterMax=iters, nout=10 ** 9)
    th = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    add_perturbation(s.arrays["T"], s.grid, **s.extra["perturbation"])
    th.T.copy_(from_numpy(s.arrays["T"], dev)); th.H.copy_(from_numpy(s.arrays["H"], dev))
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-300)
    def run(k):
        return jr.heatdiffusion_PT_(th, pt, s.flow_bcs, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=k, nout=10 ** 9, verbose=False))
    el, r = timed(run, iters)
    print(json.dumps(dict(config="thermal diffusion 2D PT (array form)", n=n, iters=iters, it_per_s=iters / el,
                          eff_GBps_at_144B=144.0 * n * n * iters / el / 1e9)))


if __name__ == "__main__":
    solcx(); shearband(); thermal()
