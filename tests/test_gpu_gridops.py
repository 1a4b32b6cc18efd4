"""GPU parity, grid operators either side of the solves (csrc/gridops.hip through the C ABI) vs the CPU oracle (bit-exact: the sums are written in the
reference's order on both sides), the reference's known answers (test/test_Interpolations.jl:43-65) on the device, argument errors, and
size-independent properties at 256^3."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(20260821)
F = lambda *s: np.asfortranarray(RNG.random(s))


def _dev():
    import torch
    return torch.device("cuda", torch.cuda.current_device())


def _up(a):
    from justrelax_jl_amd.arrays import from_numpy
    return from_numpy(a, _dev())


def _dn(t):
    from justrelax_jl_amd.arrays import to_numpy
    return to_numpy(t)


def test_reference_known_answers_2d(jr):
    """test_Interpolations.jl:26-65 on the device: StokesArrays(4 x 4), Vy = 10"""
    ni = (4, 4)
    stokes = jr.StokesArrays(jr.AMDGPUBackend, ni)
    stokes.V.Vy.fill_(10.0)
    vx, vy = jr.fzeros((5, 5), _dev(), 1.0), jr.fzeros((5, 5), _dev(), 1.0)
    jr.velocity2vertex_(vx, vy, stokes.V.Vx, stokes.V.Vy)
    assert _dn(vx)[0, 0] == 0.0 and _dn(vy)[0, 0] == 10.0
    cx, cy = jr.fzeros(ni, _dev(), 1.0), jr.fzeros(ni, _dev(), 1.0)
    jr.velocity2center_(cx, cy, stokes.V.Vx, stokes.V.Vy)
    assert _dn(cx)[0, 0] == 0.0 and _dn(cy)[0, 0] == 10.0
    ctr = F(4, 4) + 1.0
    v = jr.fzeros((5, 5), _dev())
    jr.center2vertex_harm_(v, _up(ctr))
    assert np.isclose(_dn(v)[1, 1], 4 / (1 / ctr[0, 0] + 1 / ctr[0, 1] + 1 / ctr[1, 0] + 1 / ctr[1, 1]), rtol=1e-15)      # :67-78


@pytest.mark.parametrize("ni", [(4, 4), (37, 21), (300, 130)])
def test_interpolations_2d_vs_oracle(jr, oracle, ni):
    nx, ny = ni
    Vx, Vy = F(nx + 1, ny + 2), F(nx + 2, ny + 1)
    o = oracle.velocity2vertex(Vx, Vy)
    g = [jr.fzeros((nx + 1, ny + 1), _dev()) for _ in range(2)]
    jr.velocity2vertex_(*g, _up(Vx), _up(Vy))
    for a, b in zip(o, g):
        np.testing.assert_array_equal(_dn(b), a)
    o = oracle.velocity2center(Vx, Vy)
    g = [jr.fzeros(ni, _dev()) for _ in range(2)]
    jr.velocity2center_(*g, _up(Vx), _up(Vy))
    for a, b in zip(o, g):
        np.testing.assert_array_equal(_dn(b), a)
    ctr = F(*ni) + 0.5
    v = jr.fzeros((nx + 1, ny + 1), _dev())
    jr.center2vertex_harm_(v, _up(ctr))
    np.testing.assert_array_equal(_dn(v), oracle.center2vertex_harm(ctr))
    ver = F(nx + 1, ny + 1)
    for ghost, shape in (((False, False), ni), ((True, True), (nx + 2, ny + 2)), ((True, False), (nx + 2, ny))):
        c = np.full(shape, -3.0, order="F")
        oracle.vertex2center(c, ver, ghost=ghost)
        d = jr.fzeros(shape, _dev(), -3.0)
        jr.vertex2center_(d, _up(ver), ghost_x=ghost[0], ghost_y=ghost[1])
        np.testing.assert_array_equal(_dn(d), c)


@pytest.mark.parametrize("ni", [(3, 3, 3), (17, 19, 23), (70, 33, 20)])
def test_interpolations_3d_vs_oracle(jr, oracle, ni):
    nx, ny, nz = ni
    Vx, Vy, Vz = F(nx + 1, ny + 2, nz + 2), F(nx + 2, ny + 1, nz + 2), F(nx + 2, ny + 2, nz + 1)
    dV = [_up(a) for a in (Vx, Vy, Vz)]
    for shape in (ni, (nx + 1, ny + 1, nz + 1)):            # the sizes of test_Interpolations.jl:150-164 and of the miniapps
        o = oracle.velocity2vertex(Vx, Vy, Vz, out_shape=shape)
        g = [jr.fzeros(shape, _dev()) for _ in range(3)]
        jr.velocity2vertex_(*g, *dV)
        for a, b in zip(o, g):
            np.testing.assert_array_equal(_dn(b), a)
    if ni == (3, 3, 3):                                      # the allocating form, :166-179
        g = jr.velocity2vertex(*dV)
        assert all(tuple(t.shape) == ni for t in g)
        np.testing.assert_array_equal(_dn(g[0]), oracle.velocity2vertex(Vx, Vy, Vz, out_shape=ni)[0])
    o = oracle.velocity2center(Vx, Vy, Vz)
    g = [jr.fzeros(ni, _dev()) for _ in range(3)]
    jr.velocity2center_(*g, *dV)
    for a, b in zip(o, g):
        np.testing.assert_array_equal(_dn(b), a)
    cen = [F(*ni) for _ in range(3)]
    shapes = ((nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz))
    ov = [np.full(s, -7.0, order="F") for s in shapes]
    oracle.center2vertex3d(*ov, *cen)
    gv = [jr.fzeros(s, _dev(), -7.0) for s in shapes]
    jr.center2vertex_(*gv, *[_up(c) for c in cen])
    for a, b in zip(ov, gv):
        np.testing.assert_array_equal(_dn(b), a)
    ver = F(nx + 1, ny + 1, nz + 1)
    for ghost, shape in (((False,) * 3, ni), ((True,) * 3, (nx + 2, ny + 2, nz + 2)), ((False, True, False), (nx, ny + 1, nz))):
        c = np.full(shape, -3.0, order="F")
        oracle.vertex2center(c, ver, ghost=ghost)
        d = jr.fzeros(shape, _dev(), -3.0)
        jr.vertex2center_(d, _up(ver), ghost_x=ghost[0], ghost_y=ghost[1], ghost_z=ghost[2])
        np.testing.assert_array_equal(_dn(d), c)


PHASES = [dict(eta=1e21, G=1e10, Kb=1e11, g=9.81, shear_heat=0.7, density=dict(kind="PT", rho0=3300.0, alpha=3e-5, beta=1e-11, T0=273.0)),
          dict(eta=1e19, G=2e10, Kb=1e11, shear_heat=0.2, density=dict(kind="T", rho0=2700.0, alpha=2e-5)),
          dict(eta=1e20, G=float("inf"), Kb=1e11, density=dict(kind="compressible", rho0=2900.0, beta=2e-11, P0=1e5))]


def _ratios(ni, nph):
    r = RNG.random((nph,) + tuple(ni))
    r[:, RNG.random(ni) < 0.2] = 0.0
    r[0][r.sum(0) == 0.0] = 1.0
    r /= r.sum(0)
    one = RNG.random(ni) < 0.2                   # cells of one pure phase: the isone / iszero branches of fn_ratio
    r[:, one] = 0.0
    r[1][one] = 1.0
    return np.asfortranarray(r)


@pytest.mark.parametrize("ni", [(37, 21), (20, 14, 12)])
def test_compute_rhog_vs_oracle(jr, oracle, ni):
    T, P = np.asfortranarray(RNG.random(ni) * 1500), np.asfortranarray(RNG.random(ni) * 1e9)
    pc = _ratios(ni, 3)
    rh = oracle.rheology_struct(PHASES)
    pr = SimpleNamespace(center=_up(pc))
    args = dict(T=_up(T), P=_up(P))
    out = tuple(jr.fzeros(ni, _dev(), 5.0) for _ in ni)
    jr.compute_ρg_(out, pr, PHASES, args)                                   # the tuple form fills the last component
    np.testing.assert_array_equal(_dn(out[-1]), oracle.compute_rhog(rh, T, P, pc))
    assert np.all(_dn(out[0]) == 5.0)
    one = jr.fzeros(ni, _dev())
    jr.compute_ρg_(one, PHASES[0], args)                                    # single MaterialParams
    np.testing.assert_array_equal(_dn(one), oracle.compute_rhog(oracle.rheology_struct(PHASES[:1]), T, P))
    jr.compute_ρg_(one, PHASES[0], dict(T=None, P=None))                    # no T, P: the law at T = P = 0
    assert np.all(_dn(one) == 3300.0 * (1 - 3e-5 * (0.0 - 273.0) + 1e-11 * 0.0) * 9.81)
    with pytest.raises(RuntimeError, match="no density law"):
        jr.compute_ρg_(one, dict(eta=1.0, G=1.0, Kb=1.0), args)
    Tg = np.asfortranarray(RNG.random(tuple(n + 2 for n in ni)) * 1500)     # ghosted thermal.T as args.T: read at [i, j, k] without a shift
    jr.compute_ρg_(one, PHASES[0], dict(T=_up(Tg), P=_up(P)))
    np.testing.assert_array_equal(_dn(one), oracle.compute_rhog(oracle.rheology_struct(PHASES[:1]), Tg, P))
    with pytest.raises(ValueError):
        jr.compute_ρg_(one, PHASES[0], dict(T=jr.fzeros(tuple(n - 1 for n in ni), _dev()), P=None))


def test_compute_rhog_reproduces_the_convection_setup(jr):
    """thermal_convection2D of test/test_WENO5.jl:208-214: compute_ρg!(ρg[2], rheology, (; T = thermal.T, P = stokes.P)) -- the miniapp builder's host
    arithmetic (ρ0 (1 - α T[i, j]) g on the ghosted T) equals the device operator"""
    s = jr.miniapps.thermal_convection2d(32, ar=1)
    out = jr.fzeros(s.ni, _dev())
    jr.compute_ρg_((jr.fzeros(s.ni, _dev()), out), s.extra["rheology"], dict(T=_up(s.arrays["T"]), P=_up(np.zeros(s.ni, order="F"))))
    np.testing.assert_allclose(_dn(out), s.arrays["fy"], rtol=1e-15)


@pytest.mark.parametrize("ni", [(37, 21), (20, 14, 12), (70, 17, 20)])
def test_compute_shear_heating_vs_oracle(jr, oracle, ni):
    nd = len(ni)
    stokes = jr.StokesArrays(jr.AMDGPUBackend, ni)
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, ni)
    cen = ("xx", "yy", "zz", "yz_c", "xz_c", "xy_c") if nd == 3 else ("xx", "yy", "xy_c")
    stag = ("xx", "yy", "zz", "yz", "xz", "xy") if nd == 3 else ("xx", "yy", "xy")
    tau = [np.asfortranarray((RNG.random(ni) - 0.5) * 1e7) for _ in cen]
    tau_o = [np.asfortranarray((RNG.random(ni) - 0.5) * 1e7) for _ in cen]
    eps = [np.asfortranarray((RNG.random(tuple(getattr(stokes.ε, k).shape)) - 0.5) * 1e-3) for k in stag]
    for k, a in zip(cen, tau):
        getattr(stokes.τ, k).copy_(_up(a))
    for k, a in zip(cen, tau_o):
        getattr(stokes.τ_o, k).copy_(_up(a))
    for k, a in zip(stag, eps):
        getattr(stokes.ε, k).copy_(_up(a))
    pc = _ratios(ni, 3)
    rh, chi, dt = oracle.rheology_struct(PHASES), [p.get("shear_heat", 0.0) for p in PHASES], 1.0e-4
    jr.compute_shear_heating_(thermal, stokes, SimpleNamespace(center=_up(pc)), PHASES, dt)
    ref = oracle.compute_shear_heating(tau, tau_o, eps, rh, chi, dt, phase_c=pc)
    got = _dn(thermal.shear_heating)
    np.testing.assert_array_equal(got, ref)
    assert (got == 0.0).any() and (got > 0.0).any()
    jr.compute_shear_heating_(thermal, stokes, PHASES[0], dt)                # single-phase form
    np.testing.assert_array_equal(_dn(thermal.shear_heating), oracle.compute_shear_heating(tau, tau_o, eps, oracle.rheology_struct(PHASES[:1]), chi[:1], dt))


@pytest.mark.parametrize("nd", [2, 3])
def test_compute_viscosity_single_vs_oracle(jr, oracle, nd):
    """compute_viscosity!(stokes, args, rheology::MaterialParams, cutoff; relaxation): Arrhenius table, ghosted and cell-centred T, relaxation, cutoff"""
    ni = (37, 21) if nd == 2 else (20, 14, 12)
    ph = dict(eta=5.0e20, G=70e9, Kb=float("inf"), creep=dict(kind="arrhenius", Ea=200.0e3, Va=2.6e-6, T0=1.6e3, R=8.3145, cutoff=(1.0e16, 1.0e25)))
    rh = oracle.rheology_struct([ph])
    stokes = jr.StokesArrays(jr.AMDGPUBackend, ni)
    Tg = np.asfortranarray(300.0 + RNG.random(tuple(n + 2 for n in ni)) * 3000.0)
    P = np.asfortranarray(RNG.random(ni) * 1.0e10)
    eta0 = np.asfortranarray(10.0 ** (18 + 6 * RNG.random(ni)))
    for T, nu, cut in ((Tg, 1.0, (1e16, 1e24)), (Tg, 0.01, (1e19, 1e22)), (np.asfortranarray(Tg[(slice(1, -1),) * nd]), 0.5, (-np.inf, np.inf))):
        ref = eta0.copy(order="F")
        oracle.compute_viscosity_single(ref, rh, T, P, cutoff=cut, nu=nu)
        stokes.viscosity.η.copy_(_up(eta0))
        jr.compute_viscosity_(stokes, dict(T=_up(T), P=_up(P)), ph, cut, relaxation=nu)
        np.testing.assert_allclose(_dn(stokes.viscosity.η), ref, rtol=1e-13)
        assert np.ptp(np.log10(ref)) > 1.0
    with pytest.raises(RuntimeError, match="ni .\\+ 2"):
        jr.compute_viscosity_(stokes, dict(T=jr.fzeros(tuple(n + 1 for n in ni), _dev()), P=None), ph, (0.0, 1.0))
    with pytest.raises(TypeError):
        jr.compute_viscosity_(stokes, dict(T=None, P=None), [ph, ph], (0.0, 1.0))


def test_lithostatic_pressure_vs_oracle_and_reference_formula(jr, oracle):
    """compute_lithostatic_pressure!(P, ρg, dz[, igg]) -- test/test_lithostatic_pressure2D_MPI.jl:104-126 on the device, and bit-identity with the oracle's top-down
    accumulation on random 2D / 3D columns with a constant and a per-cell height"""
    from test_oracle_gridops import _P_global
    nx, ny = 4, 8
    rhog = np.asfortranarray(np.tile(1 + 0.25 * np.arange(1, ny + 1)[None, :], (nx, 1)))
    P = jr.fzeros((nx, ny), _dev())
    jr.compute_lithostatic_pressure_(P, _up(rhog), 0.5)
    for j in range(1, ny + 1):
        assert np.allclose(_dn(P)[:, j - 1], _P_global(j, ny, 0.5), rtol=1e-14)
    jr.compute_lithostatic_pressure_(P, _up(rhog), 0.1 * np.arange(1, ny + 1))
    for j in range(1, ny + 1):
        assert np.allclose(_dn(P)[:, j - 1], _P_global(j, ny, None), rtol=1e-14)
    for shape in ((300, 70), (33, 20, 41)):
        r = np.asfortranarray(RNG.random(shape) * 3.0e4)
        for dz in (1.5e3, RNG.random(shape[-1]) * 2.0e3 + 100.0):
            Pd = jr.fzeros(shape, _dev())
            jr.compute_lithostatic_pressure_(Pd, _up(r), dz)
            np.testing.assert_array_equal(_dn(Pd), oracle.compute_lithostatic_pressure(r, dz))
    with pytest.raises(ValueError, match="same cells"):
        jr.compute_lithostatic_pressure_(jr.fzeros((4, 8), _dev()), jr.fzeros((4, 9), _dev()), 1.0)
    with pytest.raises(ValueError, match="one height per cell"):
        jr.compute_lithostatic_pressure_(jr.fzeros((4, 8), _dev()), jr.fzeros((4, 8), _dev()), np.ones(7))


def test_argument_errors(jr):
    d = _dev()
    with pytest.raises(AssertionError):                      # Interpolations.jl:238 @assert size(Vx_v) == size(Vy_v)
        jr.velocity2vertex_(jr.fzeros((5, 5), d), jr.fzeros((5, 4), d), jr.fzeros((5, 6), d), jr.fzeros((6, 5), d))
    with pytest.raises(RuntimeError, match="at most ni"):    # outputs larger than ni .+ 1 would read out of bounds
        jr.velocity2vertex_(jr.fzeros((6, 5), d), jr.fzeros((6, 5), d), jr.fzeros((5, 6), d), jr.fzeros((6, 5), d))
    with pytest.raises(RuntimeError, match="too small"):
        jr.vertex2center_(jr.fzeros((4, 4), d), jr.fzeros((5, 5), d), ghost_x=True)
    with pytest.raises(ValueError):
        jr.center2vertex_harm_(jr.fzeros((5, 4), d), jr.fzeros((4, 4), d))
    with pytest.raises(NotImplementedError):                 # no CPU fallback
        import torch
        jr.velocity2center_(torch.zeros(4, 4, dtype=torch.float64), torch.zeros(4, 4, dtype=torch.float64), torch.zeros(5, 6, dtype=torch.float64),
                            torch.zeros(6, 5, dtype=torch.float64))


def test_full_size_properties(jr):
    """256^3: a linear velocity field is reproduced exactly at the vertices and centres (the interpolations are means of points symmetric about the target),
    and vertex2center(center2vertex-like constants) keeps constants"""
    import torch
    n = 256
    d = _dev()
    stokes = jr.StokesArrays(jr.AMDGPUBackend, (n, n, n))
    ax = lambda m, off: torch.arange(m, dtype=torch.float64, device=d) + off
    # Vx lives at (i, j - 1/2, k - 1/2) in units of the spacing (ghost rows in y, z); f = 2 x + 3 y - z sampled there (exact in binary: halves)
    f = lambda X, Y, Z: 2.0 * X[:, None, None] + 3.0 * Y[None, :, None] - Z[None, None, :]
    stokes.V.Vx.copy_(f(ax(n + 1, 0.0), ax(n + 2, -0.5), ax(n + 2, -0.5)))
    stokes.V.Vy.copy_(f(ax(n + 2, -0.5), ax(n + 1, 0.0), ax(n + 2, -0.5)))
    stokes.V.Vz.copy_(f(ax(n + 2, -0.5), ax(n + 2, -0.5), ax(n + 1, 0.0)))
    out = [jr.fzeros((n + 1,) * 3, d) for _ in range(3)]
    jr.velocity2vertex_(*out, stokes.V.Vx, stokes.V.Vy, stokes.V.Vz)
    want = f(ax(n + 1, 0.0), ax(n + 1, 0.0), ax(n + 1, 0.0))
    for t in out:
        assert torch.equal(t, want)
    del out, want
    outc = [jr.fzeros((n,) * 3, d) for _ in range(3)]
    jr.velocity2center_(*outc, stokes.V.Vx, stokes.V.Vy, stokes.V.Vz)
    wantc = f(ax(n, 0.5), ax(n, 0.5), ax(n, 0.5))
    for t in outc:
        assert torch.equal(t, wantc)
    ver = f(ax(n + 1, 0.0), ax(n + 1, 0.0), ax(n + 1, 0.0)).permute(2, 1, 0).contiguous().permute(2, 1, 0)
    jr.vertex2center_(outc[0], ver)
    assert torch.equal(outc[0], wantc)
