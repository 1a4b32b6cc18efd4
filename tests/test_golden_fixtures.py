"""The CPU oracle against the committed fixtures of tests/golden/ (no GPU): guards the oracle -- the checker of every GPU parity
test -- against silent changes, and the fixture inputs against drift of the seeded builders."""
import numpy as np
import pytest

from _golden_io import load, rel_err, setup_from

TOL = 1e-13      # same code, same operation order: observed 0


def test_stokes3d_fixture(jr, oracle):
    from justrelax_jl_amd import checks
    inputs, outputs, meta = load("stokes3d_ve_10x8x7.npz")
    s, drift = setup_from(jr, inputs, meta)
    assert not drift, f"seeded builder no longer reproduces the stored inputs: {drift}"
    s.pt.ϵ_rel = s.pt.ϵ_abs = meta["eps"]
    assert (s.pt.r, s.pt.θ_dτ, s.pt.ηdτ) == (meta["r"], meta["theta_dtau"], meta["eta_dtau"])
    r = oracle.stokes3d_solve(s.arrays, checks.oracle_params3d(oracle, s))
    assert r["iter"] == meta["iter"] == 12
    assert np.allclose(r["err_evo1"], meta["err_evo1"], rtol=1e-12, atol=0)
    for k, ref in outputs.items():
        assert rel_err(s.arrays[k], ref) <= TOL, k


def test_stokes2d_fixture(jr, oracle):
    from justrelax_jl_amd import checks
    inputs, outputs, meta = load("stokes2d_ve_16x12.npz")
    s, drift = setup_from(jr, inputs, meta)
    assert not drift, drift
    s.pt.ϵ_rel = s.pt.ϵ_abs = meta["eps"]
    r = oracle.stokes2d_solve(s.arrays, checks.oracle_params2d(oracle, s))
    assert r["iter"] == meta["iter"] == 12
    assert np.allclose(r["err_evo1"], meta["err_evo1"], rtol=1e-12, atol=0)
    for k, ref in outputs.items():
        assert rel_err(s.arrays[k], ref) <= TOL, k


def test_thermal3d_fixture(jr, oracle):
    inputs, outputs, meta = load("thermal3d_diffusion_10x9x8.npz")
    s, drift = setup_from(jr, inputs, meta)
    assert not drift, drift
    b = s.flow_bcs
    kw = meta["builder_kwargs"]
    p = oracle.thermal_params3d(s.ni, s.grid._di["center"], s.dt, meta["eps"], no_flux=b.no_flux, constant_value=b.constant_value,
                                constant_flux=b.constant_flux, periodic=b.periodic, iterMax=kw["iterMax"], nout=kw["nout"])
    r = oracle.heatdiffusion_PT3d(s.arrays, p)
    assert list(r["iter_count"]) == meta["iter_count"] == [20, 40, 60]
    assert np.allclose(r["norm_ResT"], meta["norm_ResT"], rtol=1e-12, atol=0)
    for k, ref in outputs.items():
        assert rel_err(s.arrays[k], ref) <= TOL, k
