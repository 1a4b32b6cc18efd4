#!/usr/bin/env python3
"""Generates the fixtures tests/golden/*.npz: small seeded inputs and the outputs the CPU oracle (oracle/, a restatement of the
reference's kernels and drivers -- the Julia reference itself cannot run in the build container) produces for them.

    python tests/golden/make_golden.py

Each .npz holds `in_<field>` (every array the driver reads), `out_<field>` (state after the solve) and `meta` (JSON: sizes, spacings,
PT coefficients, boundary conditions, iteration counts, residual history).  tests/test_golden_fixtures.py checks the oracle against
them (no GPU), tests/test_gpu_golden.py checks the HIP path against them through the C ABI."""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

from __graft_entry__ import load_package  # noqa: E402

jr = load_package()
import oracle as orc  # noqa: E402
from justrelax_jl_amd import checks  # noqa: E402


def _bc(b):
    return {k: {f: bool(v) for f, v in getattr(b, k).items()} for k in ("free_slip", "no_slip", "periodic")}


def _save(name, inputs, outputs, meta):
    blob = {f"in_{k}": v for k, v in inputs.items()}
    blob.update({f"out_{k}": v for k, v in outputs.items()})
    blob["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(HERE / name, **blob)
    print(name, sum(v.nbytes for v in blob.values() if hasattr(v, "nbytes")) // 1024, "KiB")


def stokes3d():
    kw = dict(ni=(10, 8, 7), seed=20260821, iterMax=11, nout=4, bcs="slip_mix")
    s = jr.miniapps.random_fields3d(kw["ni"], seed=kw["seed"], iterMax=kw["iterMax"], nout=kw["nout"], bcs=kw["bcs"])
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    inputs = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r = orc.stokes3d_solve(s.arrays, checks.oracle_params3d(orc, s))
    state = ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "toxx", "toyy", "tozz", "toyz", "toxz", "toxy", "Rx", "Ry", "Rz", "RP", "divV")
    meta = dict(kind="3D visco-elastic Stokes, jrx_stokes3d_solve (Stokes3D.jl:25-186)", builder="random_fields3d", builder_kwargs=kw,
                eps=1e-30, dt=s.dt, li=s.extra["li"], r=s.pt.r, theta_dtau=s.pt.θ_dτ, eta_dtau=s.pt.ηdτ, bcs=_bc(s.flow_bcs),
                iter=int(r["iter"]), err_evo1=[float(x) for x in r["err_evo1"]])
    _save("stokes3d_ve_10x8x7.npz", inputs, {k: s.arrays[k] for k in state}, meta)


def stokes2d():
    kw = dict(ni=(16, 12), seed=20260821, iterMax=11, nout=4, bcs="free_slip")
    s = jr.miniapps.random_fields2d(kw["ni"], seed=kw["seed"], iterMax=kw["iterMax"], nout=kw["nout"], bcs=kw["bcs"])
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    inputs = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r = orc.stokes2d_solve(s.arrays, checks.oracle_params2d(orc, s))
    state = ("P", "Vx", "Vy", "txx", "tyy", "txy", "toxx", "toyy", "toxy", "Rx", "Ry", "RP", "divV")
    meta = dict(kind="2D visco-elastic Stokes, jrx_stokes2d_solve (Stokes2D.jl:181-325)", builder="random_fields2d", builder_kwargs=kw,
                eps=1e-30, dt=s.dt, r=s.pt.r, theta_dtau=s.pt.θ_dτ, eta_dtau=s.pt.ηdτ, bcs=_bc(s.flow_bcs),
                iter=int(r["iter"]), err_evo1=[float(x) for x in r["err_evo1"]])
    _save("stokes2d_ve_16x12.npz", inputs, {k: s.arrays[k] for k in state}, meta)


def thermal3d():
    kw = dict(ni=(10, 9, 8), iterMax=60, nout=20)
    s = jr.miniapps.diffusion3d(kw["ni"], iterMax=kw["iterMax"], nout=kw["nout"])
    b = s.flow_bcs
    p = orc.thermal_params3d(s.ni, s.grid._di["center"], s.dt, 1e-30, no_flux=b.no_flux, constant_value=b.constant_value,
                             constant_flux=b.constant_flux, periodic=b.periodic, iterMax=kw["iterMax"], nout=kw["nout"])
    inputs = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r = orc.heatdiffusion_PT3d(s.arrays, p)
    state = ("T", "Told", "dT", "qTx", "qTy", "qTz", "qTx2", "qTy2", "qTz2", "ResT")
    meta = dict(kind="3D PT heat diffusion, array-coefficient form, jrx_heatdiffusion_PT3d (DiffusionPT_solver.jl:34-149)", builder="diffusion3d",
                builder_kwargs=kw, eps=1e-30, dt=s.dt, iter_count=[int(x) for x in r["iter_count"]], norm_ResT=[float(x) for x in r["norm_ResT"]])
    _save("thermal3d_diffusion_10x9x8.npz", inputs, {k: s.arrays[k] for k in state if k in s.arrays}, meta)


if __name__ == "__main__":
    stokes3d()
    stokes2d()
    thermal3d()
