#!/usr/bin/env python3
"""it/s of the 3D paths next to the headline one on one GPU: 3D multiphase VEP shear band (Stokes3D.jl:447-668) and 3D PT heat
diffusion.  Fixed iteration counts (convergence disabled); prints one JSON line per case.  usage: bench3d_extra.py [n_vep] [n_thermal] [KEY=INT ...]"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd.arrays import from_numpy

dev = torch.device("cuda", 0)


def timed(fn, warm, iters):
    fn(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0), r


def shearband3d(n=256, iters=100, soft=False, creep=False):
    s = jr.miniapps.shearband3d(n, iterMax=iters - 1, nout=10 ** 9)
    if soft:      # a softening law on C and phi of the matrix phase: the yield function then reads EII_pl
        ph = [dict(p) for p in s.extra["phases"]]
        ph[0].update(softening_C=dict(kind="linear", min=0.5 * ph[0]["C"], max=ph[0]["C"], lo=0.0, hi=0.1),
                     softening_phi=dict(kind="nonlinear", xi0=30.0, Delta=10.0, mu=0.2, sigma=0.1))
        s.extra["phases"] = ph
    if creep:     # power-law (dislocation) creep on both phases: update_viscosity_τII! then reads the stress tensor (normals + 12 gathered edge values) every iteration
        ph = [dict(p) for p in s.extra["phases"]]
        ph[0].update(creep=dict(kind="dislocation", A=0.5, n=3.0, E=1.0, V=0.1, R=1.0))
        ph[1].update(creep=dict(kind="dislocation", A=2.0, n=3.3, E=0.6, V=0.0, R=1.0, apparatus="Invariant"))
        s.extra["phases"] = ph
        s.kwargs["viscosity_cutoff"] = (1e-2, 1e2)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in dict(Vx="V.Vx", Vy="V.Vy", Vz="V.Vz", eta="viscosity.η").items():
        o = st
        for p in path.split("."):
            o = getattr(o, p)
        o.copy_(from_numpy(s.arrays[k], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    for k, name in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
        getattr(pr, name).copy_(from_numpy(s.arrays[k], dev))
    del s.arrays
    ρg = tuple(jr.fzeros(s.ni, dev) for _ in range(3))
    args = dict(T=jr.fzeros(tuple(n_ + 2 for n_ in s.ni), dev, 1.1), P=st.P) if creep else None
    def run(k):
        kw = dict(iterMax=k - 1, nout=10 ** 9, verbose=False)
        if creep:
            kw.update(viscosity_cutoff=(1e-2, 1e2), viscosity_relaxation=0.1)
        return jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], args, s.dt, None, kwargs=kw)
    el, r = timed(run, 5, iters)
    cells = float(np.prod(s.ni))
    # as written: stress kernel reads ~70 array values per cell (3 edge families + centre), pressure/strain 14, viscosity 2, velocity 17, + phase arrays
    print(json.dumps(dict(config="shear band 3D multiphase VEP" + (" with softening laws" if soft else "") + (" with power-law creep" if creep else ""), n=n, iters=r.iter, it_per_s=r.iter / el, ms_per_it=el / r.iter * 1e3,
                          Mcell_updates_per_s=cells * r.iter / el / 1e6)))


def thermal3d(n=256, iters=400):
    s = jr.miniapps.diffusion3d(n, iterMax=iters, nout=10 ** 9)
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    thermal.T.copy_(from_numpy(s.arrays["T"], dev)); thermal.H.fill_(1e-6)
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-300)
    def run(k):
        jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=k, nout=10 ** 9, verbose=False))
        return k
    el, k = timed(run, 20, iters)
    cells = float(np.prod(s.ni))
    # algorithmic: flux R(T,K,θ,q(3)) W(q(3),q2(3)) = 12 ; update R(q(3),Told,ρCp,dτ_ρ,H,SH,T) W(T) = 10 -> 22 passes = 176 B/cell
    print(json.dumps(dict(config="thermal diffusion 3D (array form)", n=n, iters=k, it_per_s=k / el, ms_per_it=el / k * 1e3,
                          eff_GBps_at_176B=176.0 * cells * k / el / 1e9)))


def thermal3d_phases(n=256, iters=200):
    """phase-ratio form of the 3D heat-diffusion path (two phases, ball in the middle): update_pt_thermal_arrays! + compute_flux! + update_T! per iteration"""
    from types import SimpleNamespace
    s = jr.miniapps.diffusion3d_multiphase(n, iterMax=iters, nout=10 ** 9)
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    thermal.T.copy_(from_numpy(s.arrays["T"], dev)); thermal.H.fill_(1e-6)
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    for k, v in s.extra["phase_ratios"].items():
        getattr(pr, k).copy_(from_numpy(v, dev))
    args = SimpleNamespace(P=jr.fzeros(s.ni, dev), T=thermal.T)
    pt = jr.PTThermalCoeffs.from_phases(jr.AMDGPUBackend, s.extra["rheology"], pr, args, s.dt, s.ni, s.extra["di"], s.extra["li"], ϵ=1e-300, CFL=s.pt["CFL"])
    def run(k):
        jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], args, s.dt, s.grid, kwargs=dict(phase=pr, iterMax=k, nout=10 ** 9, verbose=False))
        return k
    el, k = timed(run, 10, iters)
    cells = float(np.prod(s.ni))
    # algorithmic: coefficients R(T, P, phase_c(2)) W(θ, dτ_ρ) = 6; flux R(T, θ, q(3), face ratios 3 x 2) W(q(3), q2(3)) = 17; update R(q(3), Told, T, P, phase_c(2), dτ_ρ, H, SH) W(T) = 12
    print(json.dumps(dict(config="thermal diffusion 3D (phase-ratio form, 2 phases)", n=n, iters=k, it_per_s=k / el, ms_per_it=el / k * 1e3,
                          eff_GBps_at_280B=280.0 * cells * k / el / 1e9)))


if __name__ == "__main__":
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    soft = "soft" in sys.argv[3:]
    for kv in (a for a in sys.argv[3:] if "=" in a):                     # library options, KEY=INT
        import ctypes as C
        from justrelax_jl_amd import _lib
        k, v = kv.split("=")
        _lib.default_handle(0).set_option(k, int(v))
        print(f"# option {k} = {v}")
    if nt > 0 and "phases" in sys.argv[3:]:
        thermal3d_phases(nt)
    elif nt > 0:
        thermal3d(nt)
    if nv > 0:
        shearband3d(nv, soft=soft, creep="creep" in sys.argv[3:])
