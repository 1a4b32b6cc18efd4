"""CPU tests of the material laws the round-2 rheology table adds to the oracle (density, softening) and of the host-side table builders.
The GeoParams forms are ASSUMED (oracle/material.h); what is checked here is that oracle and product tables are built identically and
that the laws have the documented limits."""
import ctypes as C
import math

import numpy as np
import pytest


def _phases():
    return [dict(eta=1.0, G=1.0, Kb=4.0, C=1.8, phi_deg=30.0, psi_deg=3.0, eta_vp=1e-2, g=9.81,
                 density=dict(kind="PT", rho0=3300.0, alpha=3e-5, beta=1e-11, T0=273.0, P0=1e5),
                 softening_C=dict(kind="linear", min=0.5, max=1.8, lo=0.1, hi=0.5),
                 softening_phi=dict(kind="nonlinear", xi0=30.0, Delta=10.0, mu=1.0, sigma=0.5)),
            dict(eta=0.1, G=0.5, Kb=float("inf"), density=dict(kind="constant", rho0=2700.0),
                 creep=dict(kind="arrhenius", Ea=200e3, Va=2.6e-6, T0=1.6e3, R=8.3145, cutoff=(1e16, 1e25)))]


def test_product_and_oracle_tables_are_identical(jr, oracle):
    from justrelax_jl_amd import _lib, stokes
    a, b = stokes.rheology_table(_phases()), oracle.rheology_struct(_phases())
    assert C.sizeof(a) == C.sizeof(b) == C.sizeof(_lib.Rheology)
    assert bytes(a) == bytes(b)
    assert [f[0] for f in _lib.Rheology._fields_] == [f[0] for f in oracle.Rheology._fields_]
    assert a.has_density == 1 and a.gravity == 9.81 and a.rho_kind[0] == 1 and a.rho_kind[1] == 0
    assert a.softC_kind[0] == 1 and a.softphi_kind[0] == 2 and a.visc_kind[1] == 1


def test_density_in_the_vep_driver(jr, oracle):
    """one VEP solve with has_density: ρg_y = Σ ratio ρ_phase(T, P) g with the PT_Density / ConstantDensity forms"""
    from test_oracle_golden import _vep_params
    s = jr.miniapps.shearband2d(12, iterMax=10, nout=5)
    s.kwargs.update(iterMin=1)
    T = np.asfortranarray(np.random.default_rng(0).uniform(300.0, 1600.0, size=s.ni))
    s.arrays["T"] = T
    phases = [dict(s.extra["phases"][0], g=9.81, density=dict(kind="PT", rho0=3300.0, alpha=3e-5, beta=0.0, T0=273.0)),
              dict(s.extra["phases"][1], density=dict(kind="constant", rho0=2700.0))]
    r = oracle.stokes2d_vep_solve(s.arrays, oracle.rheology_struct(phases), _vep_params(oracle, s, iterMin=1))
    ph = s.arrays["phase_c"]
    want = (ph[0] * 3300.0 * (1.0 - 3e-5 * (T - 273.0)) + ph[1] * 2700.0) * 9.81
    assert np.allclose(s.arrays["fy"], want, rtol=1e-14)
    assert r["iter"] >= 2


def test_softening_laws_limits(oracle):
    """LinearSoftening: max below lo, min above hi, linear in between; NonLinearSoftening: ξ₀ far below μ, ξ₀ - Δ far above"""
    L = oracle.lib()
    L.orc_soften.restype = C.c_double
    L.orc_soften.argtypes = [C.c_int32] + [C.c_double] * 6
    lin = lambda e: L.orc_soften(1, 0.5, 1.8, 0.1, 0.5, e, 1.8)
    assert lin(0.0) == 1.8 and lin(0.1) == 1.8 and lin(0.5) == 0.5 and lin(3.0) == 0.5
    assert lin(0.3) == pytest.approx(1.15, rel=1e-14)
    non = lambda e: L.orc_soften(2, 30.0, 10.0, 1.0, 0.5, e, 30.0)
    assert non(-50.0) == pytest.approx(30.0, abs=1e-12) and non(50.0) == pytest.approx(20.0, abs=1e-12)
    assert non(1.0) == pytest.approx(30.0 - 5.0 * math.erfc(0.0), rel=1e-14) == pytest.approx(25.0)
    assert L.orc_soften(0, 1.0, 2.0, 3.0, 4.0, 0.7, 42.0) == 42.0          # NoSoftening


def test_shearband_with_softening_yields_earlier(jr, oracle):
    """The shear-band script run into the plastic stage (unpinned by the reference, whose softening test stops in the elastic stage):
    with a cohesion that softens with accumulated plastic strain the maximum stress ends lower than without."""
    from test_oracle_golden import _vep_params
    outs = {}
    for soft in (False, True):
        s = jr.miniapps.shearband2d(16, nout=100)
        phases = [dict(p) for p in s.extra["phases"]]
        if soft:
            for p in phases:
                p["softening_C"] = dict(kind="linear", min=0.5 * p["C"], max=p["C"], lo=0.0, hi=0.05)
        rh, prm = oracle.rheology_struct(phases), _vep_params(oracle, s)
        for _ in range(12):
            r = oracle.stokes2d_vep_solve(s.arrays, rh, prm)
        outs[soft] = (float(s.arrays["tII"].max()), float(s.arrays["EII_pl"].max()), r["err_evo1"][-1])
    assert outs[False][1] > 0 and outs[True][1] > 0                 # both runs yield
    assert outs[True][0] < outs[False][0] - 1e-2                    # the softened run carries less stress
    assert outs[False][2] < 1e-5 and outs[True][2] < 1e-5


def test_weno5_stokes_problem_converges_through_the_single_phase_driver(jr, oracle):
    """test/test_WENO5.jl:226-291 drives solve!(..., rheology::MaterialParams, args, dt, igg) (Stokes2D.jl:345-557) and asserts
    iters.err_evo1[end] < 5e-4.  The Stokes problem of that script (32 x 32 cells, aspect ratio 8, its tolerances and iteration budget)
    through the oracle's restatement of that driver; at aspect ratio 8 no cell centre falls inside the 150-km anomaly, so the state is
    lithostatic; at aspect ratio 1 the anomaly is resolved and drives convection at cm/yr."""
    yr = 3600 * 24 * 365.25
    for ar, moving in ((8, False), (1, True)):
        s = jr.miniapps.thermal_convection2d(32, ar=ar)
        pt, b = s.pt, s.flow_bcs
        p = oracle.vep_params2d(s.ni, s.grid._di["center"], s.dt, dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                                free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, iterMax=s.kwargs["iterMax"], nout=s.kwargs["nout"],
                                stag_mode=1, cutoff=s.kwargs["viscosity_cutoff"], T_ghosted=True)
        r = oracle.stokes2d_nonlinear_solve(s.arrays, oracle.rheology_struct([s.extra["rheology"]]), p)
        assert r["err_evo1"][-1] < 5.0e-4 and r["iter"] < s.kwargs["iterMax"]
        vmax = max(np.abs(s.arrays["Vx"]).max(), np.abs(s.arrays["Vy"]).max()) * yr * 100          # cm / yr
        assert (0.1 < vmax < 100.0) if moving else vmax < 1e-6
        assert 1e16 <= s.arrays["eta"].min() and s.arrays["eta"].max() <= 1e24


def test_reference_rheology_helper_known_answers(jr, oracle):
    """test/test_rheology.jl:57-64,103-116,123-155,160-172: the material helpers the path calls, at the values the reference's test asserts --
    get_α (0 for ConstantDensity, α for T_Density / PT_Density: the thermal expansivity adiabatic_heating! uses), get_shear_modulus (Inf when the
    material has no elasticity), compute_ρCp = ρ Cp and the diffusivity k / (ρ Cp) behind PTThermalCoeffs, compute_buoyancy = ρ g"""
    import json
    from pathlib import Path
    ka = json.loads((Path(__file__).parent / "golden" / "reference_known_answers.json").read_text())["rheology_helpers"]
    # get_α through adiabatic_heating!: A = (P - P0) α / dt with P - P0 = 1, dt = 1
    P, P0 = np.ones(3), np.zeros(3)
    for kind, want in (("constant", ka["get_alpha"]["constant"]), ("T", ka["get_alpha"]["T_alpha_3e-5"]), ("PT", ka["get_alpha"]["PT_alpha_3e-5"])):
        m = oracle.thermal_phases([dict(k=3.0, Cp=1e3, density=dict(kind=kind, rho0=3e3, alpha=3.0e-5))], 1.0, 1.0)
        A = np.zeros(3)
        oracle.adiabatic_heating(A, P, P0, m, None, 1.0)
        assert np.all(A == want), kind
    # get_shear_modulus / get_bulk_modulus: the table's defaults are the reference's Inf fallbacks
    from justrelax_jl_amd.stokes import rheology_table
    t = rheology_table([dict(eta=1.0e21, G=ka["shear_modulus"]["G"], Kb=ka["shear_modulus"]["Kb"]), dict(eta=1.0e21, G=float("inf"), Kb=float("inf"))])
    assert (t.G[0], t.Kb[0]) == (1.0e10, 5.0e10) and math.isinf(t.G[1]) and math.isinf(t.Kb[1])
    # compute_ρCp, compute_diffusivity: the PT coefficients of a uniform material are those of K = k, ρCp = ρ Cp
    th = ka["thermal"]
    assert th["rhoCp"] == th["rho"] * th["Cp"] and th["diffusivity"] == pytest.approx(th["k"] / th["rhoCp"], rel=1e-15)
    # compute_buoyancy = ρ g
    b = ka["buoyancy"]
    rh = oracle.rheology_struct([dict(eta=1.0e21, G=float("inf"), Kb=float("inf"), g=b["g"], density=dict(kind="constant", rho0=b["rho"]))])
    out = oracle.compute_rhog(rh, np.full((3, 3), 300.0, order="F"), np.zeros((3, 3), order="F"))
    assert np.all(out == pytest.approx(b["rho_g"], rel=1e-15))
