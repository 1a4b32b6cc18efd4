"""CPU: the oracle's grid operators (oracle/gridops.c) against the reference's own assertions -- test/test_Interpolations.jl:43-65 (2D known answers),
:150-209 (3D formulas on random inputs, every output), center2vertex_harm! :67-78 -- and against closed forms for compute_ρg! / compute_shear_heating!."""
import numpy as np

RNG = np.random.default_rng(20260821)
F = lambda *s: np.asfortranarray(RNG.random(s))


def test_interpolations_2d_known_answers(oracle):
    """test_Interpolations.jl:43-65: Vx = 0, Vy = 10 on a 4 x 4 grid"""
    nx = ny = 4
    Vx, Vy = np.zeros((nx + 1, ny + 2), order="F"), np.full((nx + 2, ny + 1), 10.0, order="F")
    vx, vy = oracle.velocity2vertex(Vx, Vy)
    assert vx.shape == (nx + 1, ny + 1) and vx[0, 0] == 0.0 and vy[0, 0] == 10.0
    cx, cy = oracle.velocity2center(Vx, Vy)
    assert cx.shape == (nx, ny) and cx[0, 0] == 0.0 and cy[0, 0] == 10.0
    Vx, Vy = F(nx + 1, ny + 2), F(nx + 2, ny + 1)
    vx, vy = oracle.velocity2vertex(Vx, Vy)
    np.testing.assert_array_equal(vx, (Vx[:, :-1] + Vx[:, 1:]) / 2)          # Interpolations.jl:244-249
    np.testing.assert_array_equal(vy, (Vy[:-1, :] + Vy[1:, :]) / 2)
    cx, cy = oracle.velocity2center(Vx, Vy)
    np.testing.assert_array_equal(cx, (Vx[:-1, 1:-1] + Vx[1:, 1:-1]) / 2)    # :285-289
    np.testing.assert_array_equal(cy, (Vy[1:-1, :-1] + Vy[1:-1, 1:]) / 2)


def test_center2vertex_harm(oracle):
    """test_Interpolations.jl:67-78"""
    ctr = F(4, 4) + 1.0
    v = oracle.center2vertex_harm(ctr)
    assert np.isclose(v[1, 1], 4 / (1 / ctr[0, 0] + 1 / ctr[0, 1] + 1 / ctr[1, 0] + 1 / ctr[1, 1]), rtol=1e-15)
    assert v[0, 0] == 4 / (4 / ctr[0, 0]) and np.isclose(v[4, 2], 4 / (2 / ctr[3, 1] + 2 / ctr[3, 2]), rtol=1e-15)    # clamped stencils on the boundary


def test_interpolations_3d_formulas(oracle):
    """test_Interpolations.jl:150-209, every (i, j, k)"""
    n = 3
    Vx, Vy, Vz = F(n + 1, n + 2, n + 2), F(n + 2, n + 1, n + 2), F(n + 2, n + 2, n + 1)
    a, b, c = oracle.velocity2vertex(Vx, Vy, Vz, out_shape=(n, n, n))
    ca, cb, cc = oracle.velocity2center(Vx, Vy, Vz)
    for k in range(n):
        for j in range(n):
            for i in range(n):
                assert np.isclose(a[i, j, k], 0.25 * (Vx[i, j, k] + Vx[i, j + 1, k] + Vx[i, j, k + 1] + Vx[i, j + 1, k + 1]), rtol=1e-15)
                assert np.isclose(b[i, j, k], 0.25 * (Vy[i, j, k] + Vy[i + 1, j, k] + Vy[i, j, k + 1] + Vy[i + 1, j, k + 1]), rtol=1e-15)
                assert np.isclose(c[i, j, k], 0.25 * (Vz[i, j, k] + Vz[i, j + 1, k] + Vz[i + 1, j, k] + Vz[i + 1, j + 1, k]), rtol=1e-15)
                assert np.isclose(ca[i, j, k], (Vx[i, j + 1, k + 1] + Vx[i + 1, j + 1, k + 1]) / 2, rtol=1e-15)
                assert np.isclose(cb[i, j, k], (Vy[i + 1, j, k + 1] + Vy[i + 1, j + 1, k + 1]) / 2, rtol=1e-15)
                assert np.isclose(cc[i, j, k], (Vz[i + 1, j + 1, k] + Vz[i + 1, j + 1, k + 1]) / 2, rtol=1e-15)
    full = oracle.velocity2vertex(Vx, Vy, Vz)                                     # the miniapps' ni .+ 1 outputs
    assert full[0].shape == (n + 1,) * 3 and np.array_equal(full[0][:n, :n, :n], a)
    cyz, cxz, cxy = F(n, n, n), F(n, n, n), F(n, n, n)
    vyz, vxz, vxy = np.full((n, n + 1, n + 1), -7.0, order="F"), np.full((n + 1, n, n + 1), -7.0, order="F"), np.full((n + 1, n + 1, n), -7.0, order="F")
    oracle.center2vertex3d(vyz, vxz, vxy, cyz, cxz, cxy)
    i = j = k = 0                                                                  # test_Interpolations.jl:203-207
    assert np.isclose(vyz[i, j + 1, k + 1], 0.25 * (cyz[i, j, k] + cyz[i, j + 1, k] + cyz[i, j, k + 1] + cyz[i, j + 1, k + 1]), rtol=1e-15)
    assert np.isclose(vxz[i + 1, j, k + 1], 0.25 * (cxz[i, j, k] + cxz[i + 1, j, k] + cxz[i, j, k + 1] + cxz[i + 1, j, k + 1]), rtol=1e-15)
    assert np.isclose(vxy[i + 1, j + 1, k], 0.25 * (cxy[i, j, k] + cxy[i + 1, j, k] + cxy[i, j + 1, k] + cxy[i + 1, j + 1, k]), rtol=1e-15)
    assert np.all(vyz[:, 0, :] == -7.0) and np.all(vyz[:, :, n] == -7.0) and np.all(vxy[0] == -7.0) and np.all(vxz[n] == -7.0)   # boundary edges untouched
    assert np.all(vyz[:, 1:n, 1:n] != -7.0) and np.all(vxz[1:n, :, 1:n] != -7.0) and np.all(vxy[1:n, 1:n, :] != -7.0)


def test_vertex2center(oracle):
    """Interpolations.jl:72-96 incl. the ghost offsets"""
    v = F(6, 5)
    c = np.zeros((5, 4), order="F")
    oracle.vertex2center(c, v)
    np.testing.assert_allclose(c, 0.25 * (v[:-1, :-1] + v[1:, :-1] + v[:-1, 1:] + v[1:, 1:]), rtol=1e-15)
    cg = np.full((7, 6), 3.0, order="F")
    oracle.vertex2center(cg, v, ghost=(True, True))
    assert np.array_equal(cg[1:6, 1:5], c) and np.all(cg[0] == 3.0) and np.all(cg[:, 0] == 3.0) and np.all(cg[6] == 3.0)
    v3 = F(4, 5, 6)
    c3 = np.zeros((3, 4, 6), order="F")
    oracle.vertex2center(c3, v3, ghost=(False, False, True))
    ref = sum(v3[a:a + 3, b:b + 4, d:d + 5] for a in (0, 1) for b in (0, 1) for d in (0, 1)) * 0.125
    np.testing.assert_allclose(c3[:, :, 1:], ref, rtol=1e-15)
    assert np.all(c3[:, :, 0] == 0.0)


def test_compute_rhog_and_shear_heating(oracle):
    ni = (7, 5)
    phases = [dict(eta=1e21, G=1e10, Kb=1e11, g=9.81, density=dict(kind="PT", rho0=3300.0, alpha=3e-5, beta=1e-11, T0=273.0)),
              dict(eta=1e19, G=2e10, Kb=1e11, density=dict(kind="constant", rho0=2700.0))]
    rh = oracle.rheology_struct(phases)
    T, P = np.asfortranarray(RNG.random(ni) * 1500), np.asfortranarray(RNG.random(ni) * 1e9)
    r = RNG.random(ni)
    r[0, 0], r[1, 1] = 0.0, 1.0
    pc = np.asfortranarray(np.stack([r, 1 - r]))
    d0 = 3300.0 * (1 - 3e-5 * (T - 273.0) + 1e-11 * P)
    np.testing.assert_allclose(oracle.compute_rhog(rh, T, P), d0 * 9.81, rtol=1e-14)                      # BuoyancyForces.jl:17-21
    np.testing.assert_allclose(oracle.compute_rhog(rh, T, P, pc), (d0 * r + 2700.0 * (1 - r)) * 9.81, rtol=1e-14)   # :50-54, fn_ratio
    # a ghosted thermal.T as args.T is read at [i, j] without the shift to the centres (getindex_NamedTuple; test/test_WENO5.jl:208-214)
    Tg = np.asfortranarray(RNG.random((ni[0] + 2, ni[1] + 2)) * 1500)
    np.testing.assert_allclose(oracle.compute_rhog(rh, Tg, P), 3300.0 * (1 - 3e-5 * (Tg[:ni[0], :ni[1]] - 273.0) + 1e-11 * P) * 9.81, rtol=1e-14)
    # shear heating, 2D: Χ (τxx (εxx - εel_xx) + τyy (...) + 2 τxy (av(εxy) - εel_xy)), clipped at 0
    nx, ny = ni
    tau, tau_o = [F(*ni) - 0.5 for _ in range(3)], [F(*ni) - 0.5 for _ in range(3)]
    eps = [F(*ni) - 0.5, F(*ni) - 0.5, F(nx + 1, ny + 1) - 0.5]
    for t in tau + tau_o:
        t *= 1e7
    dt, chi = 1e-3, (0.7, 0.2)
    exy = (eps[2][:-1, :-1] + eps[2][1:, :-1] + eps[2][:-1, 1:] + eps[2][1:, 1:]) / 4
    def H(G, X):
        eel = [0.5 * (t - to) / (G * dt) for t, to in zip(tau, tau_o)]
        return np.maximum(0.0, X * (tau[0] * (eps[0] - eel[0]) + tau[1] * (eps[1] - eel[1]) + 2 * tau[2] * (exy - eel[2])))
    out = oracle.compute_shear_heating(tau, tau_o, eps, rh, chi, dt)
    np.testing.assert_allclose(out, H(1e10, 0.7), rtol=1e-12, atol=1e-3)
    assert (out == 0.0).any() and (out > 0.0).any()
    out = oracle.compute_shear_heating(tau, tau_o, eps, rh, chi, dt, phase_c=pc)
    np.testing.assert_allclose(out, H(1e10 * r + 2e10 * (1 - r), 0.7 * r + 0.2 * (1 - r)), rtol=1e-12, atol=1e-3)
    # 3D: the same contraction over six components
    n3 = (4, 5, 3)
    tau3, tau3o = [F(*n3) - 0.5 for _ in range(6)], [F(*n3) - 0.5 for _ in range(6)]
    nx, ny, nz = n3
    eps3 = [F(*n3) - 0.5 for _ in range(3)] + [F(nx, ny + 1, nz + 1) - 0.5, F(nx + 1, ny, nz + 1) - 0.5, F(nx + 1, ny + 1, nz) - 0.5]
    av = [*eps3[:3],
          0.25 * (eps3[3][:, :-1, :-1] + eps3[3][:, 1:, :-1] + eps3[3][:, :-1, 1:] + eps3[3][:, 1:, 1:]),
          0.25 * (eps3[4][:-1, :, :-1] + eps3[4][1:, :, :-1] + eps3[4][:-1, :, 1:] + eps3[4][1:, :, 1:]),
          0.25 * (eps3[5][:-1, :-1, :] + eps3[5][1:, :-1, :] + eps3[5][:-1, 1:, :] + eps3[5][1:, 1:, :])]
    ref = sum((1 if q < 3 else 2) * tau3[q] * (av[q] - 0.5 * (tau3[q] - tau3o[q]) / (1e10 * 2.0)) for q in range(6))
    out = oracle.compute_shear_heating(tau3, tau3o, eps3, rh, chi, 2.0)
    np.testing.assert_allclose(out, np.maximum(0.0, 0.7 * ref), rtol=1e-12, atol=1e-15)


def test_compute_viscosity_single(oracle):
    """compute_viscosity!(stokes, args, rheology, cutoff) of test/test_WENO5.jl:196 (Arrhenius CustomRheology :25-42, T = thermal.T read at I .+ 1): the
    miniapp builder's host arithmetic is the closed form"""
    from __graft_entry__ import load_package
    jr = load_package()
    s = jr.miniapps.thermal_convection2d(32, ar=1)
    rh = oracle.rheology_struct([s.extra["rheology"]])
    eta = np.full(s.ni, 1.0e30, order="F")
    oracle.compute_viscosity_single(eta, rh, s.arrays["T"], s.arrays["P"], cutoff=s.kwargs["viscosity_cutoff"])
    np.testing.assert_allclose(eta, s.arrays["eta"], rtol=1e-13)
    before = eta.copy(order="F")
    s.arrays["T"][...] *= 1.05
    oracle.compute_viscosity_single(eta, rh, s.arrays["T"], s.arrays["P"], cutoff=s.kwargs["viscosity_cutoff"], nu=0.25)      # continuation_linear
    full = before.copy(order="F")
    oracle.compute_viscosity_single(full, rh, s.arrays["T"], s.arrays["P"], cutoff=s.kwargs["viscosity_cutoff"])
    np.testing.assert_allclose(eta, np.clip(0.75 * before + 0.25 * full, *s.kwargs["viscosity_cutoff"]), rtol=1e-13)
    Tc = np.asfortranarray(s.arrays["T"][1:-1, 1:-1])                                    # cell-centred T gives the same numbers without the shift
    e2 = before.copy(order="F")
    oracle.compute_viscosity_single(e2, rh, Tc, s.arrays["P"], cutoff=s.kwargs["viscosity_cutoff"])
    assert np.array_equal(e2, full)


def _P_global(j, nyg, dz):
    """test/test_lithostatic_pressure2D_MPI.jl:42-49 (1-based j): pressure at cell j of a column of nyg cells, ρg_global(j) = 1 + 0.25 j, dz_global(j) = 0.1 j"""
    rg = lambda k: 1 + 0.25 * k
    if dz is None:
        return sum(rg(k) * 0.1 * k for k in range(j + 1, nyg + 1)) + rg(j) * 0.1 * j / 2
    return sum(rg(k) for k in range(j + 1, nyg + 1)) * dz + rg(j) * dz / 2


def test_lithostatic_pressure_reference_formulas(oracle):
    """compute_lithostatic_pressure!(P, ρg, dz): the single-subdomain case of test/test_lithostatic_pressure2D_MPI.jl:104-126 (nx, ny = 4, 8; constant and per-cell
    heights) and a 3D column"""
    nx, ny = 4, 8
    rhog = np.asfortranarray(np.tile(1 + 0.25 * np.arange(1, ny + 1)[None, :], (nx, 1)))
    P = oracle.compute_lithostatic_pressure(rhog, 0.5)
    for j in range(1, ny + 1):
        assert np.allclose(P[:, j - 1], _P_global(j, ny, 0.5), rtol=1e-14)
    P = oracle.compute_lithostatic_pressure(rhog, 0.1 * np.arange(1, ny + 1))
    for j in range(1, ny + 1):
        assert np.allclose(P[:, j - 1], _P_global(j, ny, None), rtol=1e-14)
    r3 = np.asfortranarray(RNG.random((5, 4, 6)) + 1.0)
    dz = RNG.random(6) + 0.5
    P3 = oracle.compute_lithostatic_pressure(r3, dz)
    w = r3 * dz[None, None, :]
    want = np.flip(np.cumsum(np.flip(w, axis=2), axis=2), axis=2) - w / 2           # Utils.jl:571
    np.testing.assert_allclose(P3, want, rtol=1e-14)
