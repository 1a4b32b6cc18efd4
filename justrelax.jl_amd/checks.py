"""Helpers shared by __graft_entry__.smoke(), bench.py and the GPU parity tests: run the same
inputs through the HIP path (C ABI) and through the CPU oracle (passed in by the caller -- this
module never imports oracle/ itself) and compare."""
from __future__ import annotations

import numpy as np

from .backend import AMDGPUBackend
from .miniapps.common import download_stokes, upload_stokes


def oracle_params3d(orc, setup, **over):
    pt = setup.pt
    kw = dict(setup.kwargs)
    kw.update(over)
    b = setup.flow_bcs
    return orc.params3d(setup.ni, setup.grid._di["center"], setup.dt,
                        dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                        iterMax=kw["iterMax"], nout=kw["nout"], free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic,
                        displacement_bcs=type(b).__name__ == "DisplacementBoundaryConditions")


def oracle_params2d(orc, setup, **over):
    pt = setup.pt
    kw = dict(setup.kwargs)
    kw.update(over)
    b = setup.flow_bcs
    return orc.params2d(setup.ni, setup.grid._di["center"], setup.dt,
                        dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                        iterMax=kw["iterMax"], nout=kw["nout"], free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic,
                        displacement_bcs=type(b).__name__ == "DisplacementBoundaryConditions")


def max_rel_diff(a: np.ndarray, b: np.ndarray) -> float:
    """max |a-b| / max(|b|, tiny) over finite entries; NaN/Inf patterns must coincide."""
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return float("inf")
    if not fa.any():
        return 0.0
    scale = max(float(np.abs(b[fb]).max()), 1e-300)
    return float(np.abs(a[fa] - b[fb]).max() / scale)


def interior_mask3d(name, shape):
    """Edge/corner ghosts of V are written racily by the reference's BC kernels and never read
    (SURVEY App. C #5): exclude points that are ghost in two or more directions."""
    m = np.ones(shape, dtype=bool)
    ghost_dims = {"Vx": (1, 2), "Vy": (0, 2), "Vz": (0, 1), "Ux": (1, 2), "Uy": (0, 2), "Uz": (0, 1)}.get(name)
    if ghost_dims is None:
        return m
    g = np.zeros(shape, dtype=int)
    for d in ghost_dims:
        idx = [slice(None)] * 3
        for e in (0, shape[d] - 1):
            idx[d] = e
            g[tuple(idx)] += 1
    return g < 2


def compare_stokes(dev: dict, ref: dict, names=None) -> dict:
    out = {}
    for k in (names or ref.keys()):
        if k not in dev or k not in ref:
            continue
        a, b = dev[k], ref[k]
        if a.ndim == 3:
            m = interior_mask3d(k, a.shape)
            a, b = a[m], b[m]
        out[k] = max_rel_diff(a, b)
    return out


def smoke_solvi3d(jr, orc, n=16, iters=50) -> float:
    """SolVi3D n^3, a fixed number of PT iterations through jrx_stokes3d_solve vs the oracle."""
    import copy
    setup = jr.miniapps.solvi3d(n, iterMax=iters - 1, nout=10)
    ref = {k: v.copy(order="F") for k, v in setup.arrays.items()}
    p = oracle_params3d(orc, setup)
    r_ref = orc.stokes3d_solve(ref, p)
    stokes, ρg, K, G = upload_stokes(setup, AMDGPUBackend)
    r = jr.solve_(stokes, setup.pt, setup.grid, setup.flow_bcs, ρg, K, G, setup.dt, None, kwargs=setup.kwargs)
    assert r.iter == r_ref["iter"], (r.iter, r_ref["iter"])
    dev = download_stokes(stokes)
    diffs = compare_stokes(dev, ref, ["P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "Rx", "Ry", "Rz", "RP"])
    worst = max(diffs.values())
    assert worst < 1e-10, diffs
    assert np.allclose(r.norm_Rx, r_ref["norm_Rx"], rtol=1e-9, atol=0)
    return worst
