"""CPU checks of the oracle's compute_τ_nonlinear! / center2vertex! restatements.

The reference holds no known-answer test for these kernels (parity unpinned, see oracle/stokes2d_vep.c), so the oracle
is checked against independent numpy restatements of the formulas of rheology/StressUpdate.jl:2-105 and
Interpolations.jl:101-114."""
import ctypes as C

import numpy as np
import pytest


def _dp(x):
    return x.ctypes.data_as(C.POINTER(C.c_double))


def _setup(jr, n=12, seed=3):
    s = jr.miniapps.shearband2d(n)
    rng = np.random.default_rng(seed)
    a = s.arrays
    for k in ("P", "exx", "eyy", "exy", "txx", "tyy", "txy", "txy_c", "toxx", "toyy", "toxy", "toxy_c"):
        a[k][...] = rng.uniform(-2.0, 2.0, size=a[k].shape)
    a["eta"][...] = 10.0 ** rng.uniform(-1.0, 0.5, size=a["eta"].shape)
    return s


@pytest.mark.parametrize("plastic", [False, True])
def test_compute_tau_nonlinear_single_phase_formulas(jr, oracle, plastic):
    s = _setup(jr)
    ph = dict(s.extra["phases"][0])
    ph.update(Kb=2.5, psi_deg=5.0)
    if not plastic:
        ph["C"] = None
    rh = oracle.rheology_struct([ph])
    pt, b = s.pt, s.flow_bcs
    p = oracle.vep_params2d(s.ni, s.grid._di["center"], s.dt, dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                            free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, stag_mode=1)
    a0 = {k: v.copy(order="F") for k, v in s.arrays.items()}
    a = {k: v.copy(order="F") for k, v in s.arrays.items()}
    nx, ny = s.ni
    lam0 = np.asfortranarray(np.random.default_rng(1).uniform(0, 0.05, size=s.ni))
    lam, theta = lam0.copy(order="F"), np.zeros(s.ni, order="F")
    f = oracle.vep2d(a)
    oracle.lib().orc_compute_tau_nonlinear2d(C.byref(f), _dp(theta), _dp(lam), C.byref(rh), C.byref(p), C.c_int32(0))

    # independent restatement (vectorised numpy)
    eta, dt, th = a0["eta"], s.dt, pt.θ_dτ
    _Gdt = 1.0 / (ph["G"] * dt)
    dtr = 1.0 / (th + (eta * _Gdt + 1.0))
    exy = a0["exy"]
    eij = [a0["exx"], a0["eyy"], (exy[:-1, :-1] + exy[1:, :-1] + exy[:-1, 1:] + exy[1:, 1:]) / 4]
    tij = [a0["txx"], a0["tyy"], a0["txy_c"]]
    toij = [a0["toxx"], a0["toyy"], a0["toxy"][:nx, :ny]]           # vertex array read at the centre index
    d = [dtr * (2.0 * eta * e + (-(t - to) * eta * _Gdt - t)) for t, to, e in zip(tij, toij, eij)]
    inv2 = lambda x, y, z: np.sqrt(0.5 * (x * x + y * y) + z * z)
    tII_tr = inv2(*[t + q for t, q in zip(tij, d)])
    if plastic:
        sinphi, cosphi, sinpsi = np.sin(np.radians(ph["phi_deg"])), np.cos(np.radians(ph["phi_deg"])), np.sin(np.radians(5.0))
        ty = np.maximum(ph["C"] * cosphi + a0["P"] * sinphi, 0.0)
        y = tII_tr > ty
        F = tII_tr - ty
        vol = ph["Kb"] * dt * sinphi * sinpsi
        lam_new = np.where(y, 0.5 * lam0 + 0.5 * F / (eta * dtr + ph.get("eta_vp", 0.0) + vol), lam0)
        ldq = [np.where(y, (t + q) * lam_new * 0.5 / tII_tr, 0.0) for t, q in zip(tij, d)]
        d = [np.where(y, q - 2.0 * dtr * eta * l, q) for q, l in zip(d, ldq)]
        assert y.any() and (~y).any()
    else:
        sinpsi, lam_new, ldq = 0.0, lam0, [np.zeros(s.ni)] * 3
    rt = 1e-13
    for k, t, q in zip(("txx", "tyy", "txy_c"), tij, d):
        assert np.allclose(a[k], t + q, rtol=rt, atol=1e-14), k
    assert np.allclose(a["tII"], inv2(*[t + q for t, q in zip(tij, d)]), rtol=rt)
    assert np.allclose(a["eta_vep"], a["tII"] * 0.5 / inv2(*eij), rtol=rt)
    assert np.allclose(lam, lam_new, rtol=rt)
    assert np.allclose(a["eplxx"], ldq[0], rtol=rt, atol=1e-16) and np.allclose(a["eplxy"][:nx, :ny], ldq[2], rtol=rt, atol=1e-16)
    assert np.array_equal(a["eplxy"][nx], a0["eplxy"][nx]) and np.array_equal(a["txy"], a0["txy"])
    assert np.allclose(theta, a0["P"] + (ph["Kb"] * dt * lam_new * sinpsi if plastic else 0.0), rtol=rt)


def test_multiphase_form_reduces_to_single_phase_on_pure_cells(jr, oracle):
    s = _setup(jr, seed=8)
    phases = [dict(ph, Kb=2.0, psi_deg=3.0) for ph in s.extra["phases"]]
    rh2, rh1 = oracle.rheology_struct(phases), oracle.rheology_struct(phases[1:])
    pt = s.pt
    p = oracle.vep_params2d(s.ni, s.grid._di["center"], s.dt, dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs), stag_mode=1)
    s.arrays["phase_c"][0], s.arrays["phase_c"][1] = 0.0, 1.0          # every cell is phase 2
    outs = []
    for rh, multi in ((rh2, 1), (rh1, 0)):
        a = {k: v.copy(order="F") for k, v in s.arrays.items()}
        lam, theta = np.full(s.ni, 0.01, order="F"), np.zeros(s.ni, order="F")
        f = oracle.vep2d(a)
        oracle.lib().orc_compute_tau_nonlinear2d(C.byref(f), _dp(theta), _dp(lam), C.byref(rh), C.byref(p), C.c_int32(multi))
        outs.append((a, lam, theta))
    for k in ("txx", "tyy", "txy_c", "tII", "eta_vep", "eplxx", "eplyy", "eplxy"):
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_center2vertex(oracle):
    rng = np.random.default_rng(0)
    nx, ny = 9, 6
    c = np.asfortranarray(rng.standard_normal((nx, ny)))
    v = np.asfortranarray(rng.standard_normal((nx + 1, ny + 1)))
    oracle.lib().orc_center2vertex2d(_dp(v), _dp(c), C.c_int64(nx), C.c_int64(ny))
    w = np.empty_like(v)
    w[1:-1, 1:-1] = (c[:-1, :-1] + c[1:, :-1] + c[:-1, 1:] + c[1:, 1:]) * 0.25
    w[0, :] = w[1, :]; w[-1, :] = w[-2, :]; w[:, 0] = w[:, 1]; w[:, -1] = w[:, -2]
    assert np.array_equal(v[1:-1, 1:-1], w[1:-1, 1:-1])
    assert np.array_equal(v, w)


def test_yield_function_gradients_and_dtau_pl_known_answers(oracle):
    """test/test_Utils.jl:399-470: DruckerPrager_regularised(C = 1, ϕ = 30, η_vp = 1e-3, Ψ = 0), G = Kb = 1, η = 1"""
    L = oracle.lib()
    L.orc_yieldfunction_phase.restype = C.c_double
    L.orc_compute_dtau_pl.restype = C.c_double
    rh = oracle.rheology_struct([dict(eta=1.0, G=1.0, Kb=1.0, C=1.0, phi_deg=30.0, psi_deg=0.0, eta_vp=1e-3)])
    one = (C.c_double * 1)(1.0)
    F_above = L.orc_yieldfunction_phase(C.byref(rh), one, C.c_double(0.0), C.c_double(5.0))
    F_below = L.orc_yieldfunction_phase(C.byref(rh), one, C.c_double(0.0), C.c_double(0.1))
    assert F_above > 0.0 and F_below < 0.0
    assert F_above == pytest.approx(5.0 - np.cos(np.radians(30.0)))                 # F = τII − C cosϕ − P sinϕ
    t = (C.c_double * 3)(1.0, -1.0, 0.5)
    dQ, dQdP, dFdP = (C.c_double * 3)(), C.c_double(), C.c_double()
    L.orc_plastic_gradients_phase2d(C.byref(rh), one, t, dQ, C.byref(dQdP), C.byref(dFdP))
    tII = np.sqrt(0.5 * (1 + 1) + 0.25)
    assert np.allclose(dQ[:], [0.5 * 1.0 / tII, -0.5 / tII, 0.5 * 0.5 / tII]) and dQdP.value == 0.0 and dFdP.value == pytest.approx(-0.5)
    tij, dtij = (C.c_double * 3)(1.0, 2.0, 0.5), (C.c_double * 3)(0.1, 0.2, 0.05)
    dpl, ldq = (C.c_double * 3)(), (C.c_double * 3)()
    lam = L.orc_compute_dtau_pl(tij, dtij, C.c_double(1.0), C.c_double(2.5), C.c_double(1e21), C.c_double(0.0), C.c_double(1e18), C.c_double(1e-22),
                                C.c_double(0.0), dpl, ldq)
    assert lam > 0.0 and lam == pytest.approx(0.5 * 1.5 / (0.1 + 1e18))
    lam2 = L.orc_compute_dtau_pl(tij, dtij, C.c_double(10.0), C.c_double(0.5), C.c_double(1e21), C.c_double(0.0), C.c_double(1e18), C.c_double(1e-22),
                                 C.c_double(0.0), dpl, ldq)
    assert lam2 == 0.0
    assert L.orc_isyielding(1, C.c_double(2.0), C.c_double(1.0)) == 1 and L.orc_isyielding(1, C.c_double(0.5), C.c_double(1.0)) == 0
    assert L.orc_isyielding(0, C.c_double(2.0), C.c_double(1.0)) == 0
