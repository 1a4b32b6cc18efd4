#!/usr/bin/env python3
"""The reference's 3D shear-heating script (test/test_shearheating3D.jl, miniapps/benchmarks/stokes3D/shear_heating) without the particles, on one MI355X:
dislocation-creep matrix with a weak inclusion under compression -> solve! -> tensor_invariant! -> compute_dt -> compute_shear_heating!.
usage: python examples/shearheating3d.py [n=32]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from __graft_entry__ import load_package

jr = load_package()
from justrelax_jl_amd.arrays import from_numpy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
s = jr.miniapps.shearheating3d(n)
stokes = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
for k, path in dict(Vx="V.Vx", Vy="V.Vy", Vz="V.Vz", P="P", eta="viscosity.η").items():
    o = stokes
    for p in path.split("."):
        o = getattr(o, p)
    o.copy_(from_numpy(s.arrays[k], dev))
pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
for k, name in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
    getattr(pr, name).copy_(from_numpy(s.arrays[k], dev))
ρg = tuple(from_numpy(s.arrays[k], dev) for k in ("fx", "fy", "fz"))
thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
thermal.T.copy_(from_numpy(s.arrays["T"], dev))
args, phases = dict(T=thermal.T, P=stokes.P), s.extra["phases"]
jr.compute_viscosity_(stokes, pr, args, phases, (-np.inf, np.inf))
r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, args, s.dt, None, kwargs=dict(s.kwargs, verbose=True))
jr.tensor_invariant_(stokes.ε)
dt = jr.compute_dt_(stokes, s.extra["di"], s.extra["dt_diff"]) * 0.1
jr.compute_shear_heating_(thermal, stokes, pr, phases, dt)
η, sh = jr.to_numpy(stokes.viscosity.η), jr.to_numpy(thermal.shear_heating)
print(f"{r.iter} PT iterations, err = {r.err_evo1[-1]:.3e};  η in [{η.min():.3e}, {η.max():.3e}] Pa s;  shear heating in [{sh.min():.3e}, {sh.max():.3e}] W/m^3;  dt = {dt:.3e} s")
