#!/usr/bin/env python3
"""jrx_field_tune on the other 3D paths: VEP shear band 3D (256^3), thermal 3D (256^3), Stokes 256^3 -- does the placement search pay there?  it/s before / after in one process.
   probe_tune_other.py [n=256] [draws=8]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib, arrays
from justrelax_jl_amd.arrays import from_numpy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)


def timed_ms(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k


def vep(h):
    s = jr.miniapps.shearband3d(n, iterMax=99, nout=10 ** 9)
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
    run = lambda k: jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=dict(iterMax=k - 1, nout=10 ** 9, verbose=False), handle=h)
    return run, (st, pr, ρg)


def thermal(h):
    s = jr.miniapps.diffusion3d(n, iterMax=400, nout=10 ** 9)
    th = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    th.T.copy_(from_numpy(s.arrays["T"], dev)); th.H.fill_(1e-6)
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-300)
    run = lambda k: jr.heatdiffusion_PT_(th, pt, s.flow_bcs, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=k, nout=10 ** 9, verbose=False), handle=h)
    return run, (th, K, ρCp, pt)


for name, build, k_probe, k_meas in (("VEP shear band 3D", vep, 20, 100), ("thermal 3D", thermal, 100, 400)):
    for rep in range(2):
        h = _lib.Handle(0)
        h.set_option("field_placement", 1); h.set_option("field_chunk_mib", 0)
        arrays.use_library_arrays(h)
        try:
            run, keep = build(h)
            run(5)
            before = timed_ms(run, k_meas)
            t0 = time.time()
            ms, kept = arrays.tune_placement(h, lambda: (run(3), timed_ms(run, k_probe))[1], draws)
            secs = time.time() - t0
            after = timed_ms(run, k_meas)
            print(f"{name} {n}^3: {1e3 / before:.1f} it/s as allocated -> {1e3 / after:.1f} it/s after {draws} draws ({kept} kept, {secs:.1f} s); ms per iteration of the draws: " + " ".join(f"{x:.3f}" for x in ms), flush=True)
            del run, keep
        finally:
            arrays.use_library_arrays(None)
            h.close()
            torch.cuda.empty_cache()
