"""CPU checks of the oracle's 3D multiphase VEP restatement (oracle/stokes3d_vep.c).

The reference's only 3D VEP test (test/test_shearband3D_MPI.jl) asserts nothing, so 3D parity is unpinned by the reference.
The restatement is anchored instead on (i) the 2D kernel it mirrors -- itself pinned by the shear-band regression scalars of
test/test_shearband2D.jl -- through a plane-strain problem that both must solve identically up to round-off, and (ii) the
analytic visco-elastic build-up the reference's 3D script prints next to its result (`solution`, test_shearband3D_MPI.jl:38)."""
import ctypes as C
import math

import numpy as np
import pytest


def _dp(x):
    return x.ctypes.data_as(C.POINTER(C.c_double))


def test_stress_kernel_reduces_to_the_2d_kernel_in_plane_strain(jr, oracle):
    orc = oracle
    n2 = 14
    s2 = jr.miniapps.shearband2d(n2)
    rng = np.random.default_rng(21)
    a2 = s2.arrays
    for k in ("P", "exx", "eyy", "exy", "txx", "tyy", "txy", "txy_c", "toxx", "toyy", "toxy", "toxy_c"):
        a2[k][...] = rng.uniform(-2.0, 2.0, size=a2[k].shape)
    a2["eta"][...] = 10.0 ** rng.uniform(-1.0, 0.5, size=a2["eta"].shape)
    for k in ("phase_c", "phase_v"):
        r = rng.uniform(0.0, 1.0, size=a2[k].shape[1:])
        r[rng.uniform(size=r.shape) < 0.3] = 0.0
        r[rng.uniform(size=r.shape) < 0.3] = 1.0
        a2[k][0], a2[k][1] = r, 1.0 - r
    phases = [dict(ph, psi_deg=4.0) for ph in s2.extra["phases"]]          # Kb = 4: exercise the dilatant terms too
    rh = orc.rheology_struct(phases)
    pt = s2.pt
    ptd = dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs)
    nx, nz = s2.ni
    ny = 3
    theta2 = np.asfortranarray(rng.uniform(-1, 1, size=s2.ni))
    lam2 = np.asfortranarray(rng.uniform(0, 0.1, size=s2.ni))
    lamv2 = np.asfortranarray(rng.uniform(0, 0.1, size=(nx + 1, nz + 1)))

    # ---- 3D problem, uniform in y: (x, y2D) -> (x, z); (xx, yy, xy)2D -> (xx, zz, xz)3D; out-of-plane members zero
    shp = orc.vep_shapes3d(nx, ny, nz, 2)
    a3 = {k: np.zeros(v, order="F") for k, v in shp.items()}
    ext = lambda A: np.asfortranarray(np.repeat(A[:, None, :], ny, axis=1))
    for k3, k2 in (("exx", "exx"), ("ezz", "eyy"), ("txx", "txx"), ("tzz", "tyy"), ("toxx", "toxx"), ("tozz", "toyy"), ("txz_c", "txy_c"),
                   ("toxz_c", "toxy_c"), ("eta", "eta"), ("exz", "exy"), ("txz", "txy"), ("toxz", "toxy")):
        a3[k3][...] = ext(a2[k2])
    for q in range(2):
        a3["phase_c"][q] = ext(a2["phase_c"][q])
        a3["phase_xz"][q] = ext(a2["phase_v"][q])
    a3["phase_yz"][0], a3["phase_xy"][0] = 1.0, 1.0
    theta3, lam3 = ext(theta2), ext(lam2)
    lamv3 = [np.zeros(shp["tyz"], order="F"), ext(lamv2), np.zeros(shp["txy"], order="F")]
    p3 = orc.vep_params3d((nx, ny, nz), (1.0, 1.0, 1.0), s2.dt, ptd)
    orc.vep3d_stress(a3, theta3, lam3, lamv3, rh, p3)

    p2 = orc.vep_params2d(s2.ni, (1.0, 1.0), s2.dt, ptd, stag_mode=1)
    f2 = orc.vep2d(a2)
    orc.lib().orc_vep2d_stress(C.byref(f2), _dp(theta2), _dp(lam2), _dp(lamv2), C.byref(rh), C.byref(p2))

    def close(A3, A2, name):
        for j in range(ny):
            assert np.allclose(A3[:, j, :], A2, rtol=1e-11, atol=1e-13), (name, j, np.abs(A3[:, j, :] - A2).max())
    for k3, k2 in (("txx", "txx"), ("tzz", "tyy"), ("txz_c", "txy_c"), ("txz", "txy"), ("tII", "tII"), ("eta_vep", "eta_vep"), ("P", "P"),
                   ("eplxx", "eplxx"), ("eplzz", "eplyy"), ("eplxz", "eplxy"), ("evol_pl", "evol_pl")):
        close(a3[k3], a2[k2], k3)
    close(lam3, lam2, "lam")
    close(lamv3[1], lamv2, "lamv_xz")
    for k in ("tyy", "tyz", "txy", "tyz_c", "txy_c", "eplyz", "eplxy"):
        assert not a3[k].any(), k                                       # out-of-plane members stay zero
    assert (lam2 > 0.1).any() or (a2["eplxx"] != 0).any()               # plastic branch exercised


def test_shearband3d_first_step_follows_the_viscoelastic_buildup(jr, oracle):
    orc = oracle
    s = jr.miniapps.shearband3d(12, iterMax=3000, nout=100)
    rh = orc.rheology_struct(s.extra["phases"])
    pt, b = s.pt, s.flow_bcs
    p = orc.vep_params3d(s.ni, s.grid._di["center"], s.dt, dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                         iterMax=3000, nout=100, free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic)
    r = orc.stokes3d_vep_solve(s.arrays, rh, p)
    assert r["err_evo1"][-1] / r["err_evo1"][0] < 1e-5 and r["iter"] < 3000
    ε, G0, η0 = s.extra["εbg"], s.extra["G0"], s.extra["η0"]
    sol = 2 * ε * η0 * (1 - math.exp(-G0 * s.dt / η0))                 # solution(ε, t, G, η), test_shearband3D_MPI.jl:38
    top = s.arrays["txx"].max()                                         # what the reference script records: maximum(stokes.τ.xx)
    assert abs(top - sol) / sol < 2e-2, (top, sol)
    assert np.abs(s.arrays["tzz"] + s.arrays["txx"]).max() < 0.05 and np.abs(s.arrays["tyy"]).max() < 0.05      # pure shear in x-z
    assert not s.arrays["eplxx"].any()                                  # below yield after the first step (τII < 1.6)
    assert np.array_equal(s.arrays["toxx"], s.arrays["txx"]) and np.array_equal(s.arrays["toxz"], s.arrays["txz"])


def test_epilogue_kernels_3d(oracle):
    orc = oracle
    rng = np.random.default_rng(0)
    nx, ny, nz = 5, 4, 3
    shp = orc.vep_shapes3d(nx, ny, nz, 1)
    A = {k: np.asfortranarray(rng.standard_normal(shp[k])) for k in ("txx", "tyy", "tzz", "tyz", "txz", "txy")}
    II = np.zeros((nx, ny, nz), order="F")
    orc.lib().orc_tensor_invariant3d(_dp(II), *[_dp(A[k]) for k in ("txx", "tyy", "tzz", "tyz", "txz", "txy")], C.c_int64(nx), C.c_int64(ny), C.c_int64(nz))
    yz, xz, xy = A["tyz"], A["txz"], A["txy"]
    m = lambda *t: sum(x * x for x in t) / 4
    want = np.sqrt(0.5 * (A["txx"] ** 2 + A["tyy"] ** 2 + A["tzz"] ** 2)
                   + m(yz[:, :-1, :-1], yz[:, 1:, :-1], yz[:, :-1, 1:], yz[:, 1:, 1:])
                   + m(xz[:-1, :, :-1], xz[1:, :, :-1], xz[:-1, :, 1:], xz[1:, :, 1:])
                   + m(xy[:-1, :-1, :], xy[1:, :-1, :], xy[:-1, 1:, :], xy[1:, 1:, :]))
    assert np.allclose(II, want, rtol=1e-14)
    c = [np.zeros((nx, ny, nz), order="F") for _ in range(3)]
    orc.lib().orc_shear2center3d(*[_dp(x) for x in c], _dp(yz), _dp(xz), _dp(xy), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz))
    assert np.allclose(c[0], 0.25 * (yz[:, :-1, :-1] + yz[:, 1:, :-1] + yz[:, :-1, 1:] + yz[:, 1:, 1:]), rtol=1e-14)
    assert np.allclose(c[2], 0.25 * (xy[:-1, :-1, :] + xy[1:, :-1, :] + xy[:-1, 1:, :] + xy[1:, 1:, :]), rtol=1e-14)
    # vorticity of a rigid rotation about y (Vx = z, Vz = -x on the un-shifted index lattice the reference uses): ωxz = 1
    Vx = np.asfortranarray(np.broadcast_to(np.arange(nz + 2.0)[None, None, :], (nx + 1, ny + 2, nz + 2)).copy())
    Vy = np.zeros((nx + 2, ny + 1, nz + 2), order="F")
    Vz = np.asfortranarray(np.broadcast_to(-np.arange(nx + 2.0)[:, None, None], (nx + 2, ny + 2, nz + 1)).copy())
    w = [np.zeros(shp[k], order="F") for k in ("tyz", "txz", "txy")]
    orc.lib().orc_compute_vorticity3d(*[_dp(x) for x in w], _dp(Vx), _dp(Vy), _dp(Vz), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz),
                                      C.c_double(1.0), C.c_double(1.0), C.c_double(1.0))
    assert np.allclose(w[1], 1.0) and not w[0].any() and not w[2].any()
