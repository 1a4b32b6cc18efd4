#!/usr/bin/env python3
"""Thermo-mechanical time loop of test/test_WENO5.jl:216-282 (thermal_convection2D, circular perturbation) through the native backend, without the particle /
WENO advection of temperature (out of scope of the library): per step solve! (single MaterialParams: update_ρg! and the Arrhenius viscosity relaxation run inside),
compute_dt, compute_shear_heating!, heatdiffusion_PT! (rheology form), velocity2vertex!, one .vtr file.
    python examples/thermal_convection2d.py [n=64] [steps=3] [outdir=thermal_convection2d_out]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from __graft_entry__ import load_package

jr = load_package()
from justrelax_jl_amd.arrays import from_numpy


def main(n=64, steps=3, outdir="thermal_convection2d_out"):
    from test_gpu_vep2d import VEP_MAP, _get
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    s = jr.miniapps.thermal_convection2d(n, ar=1, iterMax=20_000, nout=1000)
    di, li = s.extra["di"], s.extra["li"]
    stokes_rheology = dict(s.extra["rheology"], shear_heat=1.0)
    thermal_rheology = dict(k=3.0, Cp=1.2e3, rho0=3.1e3, alpha=1.5e-5, T0=0.0)
    κ = 3.0 / (1.2e3 * 3.1e3)
    dt_diff = 0.5 * min(di) ** 2 / κ / 2.01
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP_MAP.items():
        _get(st, path).copy_(from_numpy(s.arrays[k], dev))
    ρg = (from_numpy(s.arrays["fx"], dev), from_numpy(s.arrays["fy"], dev))
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    thermal.T.copy_(from_numpy(s.arrays["T"], dev))
    K, ρCp = jr.fzeros(s.ni, dev, 3.0), jr.fzeros(s.ni, dev, 1.2e3 * 3.1e3)
    pt_thermal = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, di, li, CFL=1.0e-3 / np.sqrt(2.1), ϵ=1.0e-5)
    dt, t, yr = s.dt, 0.0, 3600 * 24 * 365.25
    for it in range(1, steps + 1):
        args = dict(T=thermal.T, P=st.P)
        r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, stokes_rheology, args, dt, None, kwargs=s.kwargs)
        dt = jr.compute_dt_(st, di, dt_diff)
        jr.compute_shear_heating_(thermal, st, stokes_rheology, dt)
        rt = jr.heatdiffusion_PT_(thermal, pt_thermal, s.extra["thermal_bc"], thermal_rheology, None, dt, s.grid, kwargs=dict(iterMax=10_000, nout=100, verbose=False))
        t += dt
        Vx_v, Vy_v = jr.fzeros((n + 1, n + 1), dev), jr.fzeros((n + 1, n + 1), dev)
        jr.velocity2vertex_(Vx_v, Vy_v, st.V.Vx, st.V.Vy)
        vmax = max(float(Vx_v.abs().max()), float(Vy_v.abs().max())) * yr * 100
        print(f"step {it}: t = {t / yr / 1e6:.3f} Myr  Stokes iterations = {r.iter} (err {r.err_evo1[-1]:.2e})  heat iterations = {int(rt.iter_count[-1])}  "
              f"max |V| = {vmax:.3f} cm/yr  max shear heating = {float(thermal.shear_heating.max()):.3e} W/m^3", flush=True)
        jr.save_vtk(str(out / f"step_{it:04d}"), s.grid.xvi, s.grid.xci, {}, dict(T=jr.to_numpy(thermal.T)[1:-1, 1:-1], eta=jr.to_numpy(st.viscosity.η)),
                    (jr.to_numpy(Vx_v), jr.to_numpy(Vy_v)), t=t)
    return r


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 64, int(a[1]) if len(a) > 1 else 3, a[2] if len(a) > 2 else "thermal_convection2d_out")
