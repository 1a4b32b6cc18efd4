#!/usr/bin/env python3
"""The refined shear band of miniapps/benchmarks/stokes2D/shear_band/ShearBand2D_refined.jl through the native backend: the multiphase visco-elasto-plastic
2D solve! on a grid whose vertices are refined towards the inclusion in x (Geometry(xvi...)), time steps with the stress history carried as the script does
(tensor_invariant!, τ -> τ_o is done inside solve!), and a .vtr file per step.
    python examples/shearband2d_refined.py [n=64] [steps=5] [outdir=shearband2d_refined_out]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package

jr = load_package()


def refined(n, k=1.8):
    """vertices in [0, 1] clustered around 0.5 (the monitor-function grid of ShearBand2D_refined.jl:205-210 is replaced by a sinh map)"""
    s = np.linspace(-1.0, 1.0, n + 1)
    return (np.sinh(k * s) / np.sinh(k) + 1.0) / 2.0


def main(n=64, steps=5, outdir="shearband2d_refined_out"):
    from test_gpu_vep2d import _upload
    out = Path(outdir)
    out.mkdir(parents=True, exist_ok=True)
    s = jr.miniapps.shearband2d(n, iterMax=50_000, nout=1000, xvi=(refined(n), np.linspace(0.0, 1.0, n + 1)))
    s.kwargs.update(verbose=False)
    st, pr, ρg = _upload(jr, s)
    t = 0.0
    for it in range(1, steps + 1):
        r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
        jr.tensor_invariant_(st.ε)
        t += s.dt
        τII, εII = jr.to_numpy(st.τ.II), jr.to_numpy(st.ε.II)
        print(f"step {it}: t = {t:.3f}  PT iterations = {r.iter}  err = {r.err_evo1[-1]:.3e}  max τII = {τII.max():.5f}  max εII = {εII.max():.4f}", flush=True)
        Vx_v, Vy_v = jr.fzeros((n + 1, n + 1), st.P.device), jr.fzeros((n + 1, n + 1), st.P.device)
        jr.velocity2vertex_(Vx_v, Vy_v, st.V.Vx, st.V.Vy)
        jr.save_vtk(str(out / f"step_{it:04d}"), s.grid.xvi, s.grid.xci, {}, dict(tauII=τII, epsII=εII, P=jr.to_numpy(st.P), eta_vep=jr.to_numpy(st.viscosity.η_vep)),
                    (jr.to_numpy(Vx_v), jr.to_numpy(Vy_v)), t=t)
    return r


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 64, int(a[1]) if len(a) > 1 else 5, a[2] if len(a) > 2 else "shearband2d_refined_out")
