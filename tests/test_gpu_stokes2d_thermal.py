"""GPU parity tests: 2D visco-elastic Stokes (Stokes2D.jl:181-325) and 2D PT heat diffusion
(DiffusionPT_solver.jl) through the C ABI vs the CPU oracle.  Tolerances as in test_gpu_stokes3d.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_SWEEP, TOL_ITERS = 1e-12, 1e-9


def _cp(arrs):
    return {k: v.copy(order="F") for k, v in arrs.items()}


@pytest.fixture(scope="module")
def env(jr, oracle):
    import torch
    assert torch.cuda.is_available()
    from justrelax_jl_amd import checks, stokes, thermal
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    return dict(jr=jr, orc=oracle, ck=checks, st=stokes, th=thermal, up=upload_stokes, down=download_stokes)


@pytest.mark.parametrize("ni", [(17, 19), (64, 33), (3, 3), (130, 7)])
def test_2d_sweeps_match_reference_kernels(env, ni):
    jr, orc, ck, st = env["jr"], env["orc"], env["ck"], env["st"]
    s = jr.miniapps.random_fields2d(ni)
    ref = _cp(s.arrays)
    p = ck.oracle_params2d(orc, s)
    et = orc.compute_maxloc(ref["eta"])
    L = orc.lib()
    f = orc.fields2d(ref)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L.orc_compute_divV2d(f.divV, f.Vx, f.Vy, C.c_int64(ni[0]), C.c_int64(ni[1]), C.c_double(p._dx), C.c_double(p._dy))
    L.orc_compute_P3d(f.P, f.P0, f.RP, f.divV, f.Q, dp(et), f.K, f.G, C.c_int64(ni[0] * ni[1]), C.c_double(s.dt),
                      C.c_double(s.pt.r), C.c_double(s.pt.θ_dτ))           # ητ, not η, in the 2D driver (Stokes2D.jl:232)
    orc.call2d("orc_compute_strain_rate2d", ref, p)
    orc.call2d("orc_compute_tau2d", ref, p)
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    etd = jr.fzeros(ni, stokes.P.device)
    jr.compute_maxloc_(etd, stokes.viscosity.η)
    assert np.array_equal(jr.to_numpy(etd), et)
    st.sweep_stress_(stokes, s.pt, s.grid, K, G, s.dt, ητ=etd, diag=True)
    dev = env["down"](stokes)
    d = ck.compare_stokes(dev, ref, ["divV", "P", "RP", "exx", "eyy", "exy", "txx", "tyy", "txy"])
    assert max(d.values()) <= TOL_SWEEP, d
    # velocity sweep + compute_Res!
    orc.call2d("orc_compute_V2d", ref, p, dp(et))
    orc.call2d("orc_velocity2displacement2d", ref, p)
    orc.call2d("orc_compute_Res2d", ref, p)
    st.compute_Res_(stokes, s.pt, s.grid, ρg)
    st.sweep_velocity_(stokes, s.pt, s.grid, ρg, etd, s.dt, diag=True)
    dev = env["down"](stokes)
    d = ck.compare_stokes(dev, ref, ["Vx", "Vy", "Ux", "Uy", "Rx", "Ry"])
    assert max(d.values()) <= TOL_SWEEP, d
    assert np.allclose(st.residual_sumsq(stokes, s.pt, s.grid), orc.residual_sumsq2d(ref, p), rtol=1e-13, atol=0)


@pytest.mark.parametrize("kind", ["free_slip", "no_slip", "periodic"])
def test_flow_bcs2d(env, kind):
    jr, orc = env["jr"], env["orc"]
    ni = (7, 5)
    s = jr.miniapps.random_fields2d(ni, bcs=kind, seed=3)
    ref = _cp(s.arrays)
    b = s.flow_bcs
    orc.flow_bcs2d(ref["Vx"], ref["Vy"], ni, b.free_slip, b.no_slip, b.periodic)
    stokes, *_ = env["up"](s, jr.AMDGPUBackend)
    jr.flow_bcs_(stokes, b)
    dev = env["down"](stokes)
    assert np.array_equal(dev["Vx"], ref["Vx"]) and np.array_equal(dev["Vy"], ref["Vy"])


@pytest.mark.parametrize("ni,bcs", [((17, 19), "free_slip"), ((32, 32), "no_slip"), ((40, 9), "periodic")])
def test_2d_solve_matches_oracle(env, ni, bcs):
    jr, orc, ck = env["jr"], env["orc"], env["ck"]
    s = jr.miniapps.random_fields2d(ni, bcs=bcs, iterMax=20, nout=5)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    ref = _cp(s.arrays)
    r_ref = orc.stokes2d_solve(ref, ck.oracle_params2d(orc, s))
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)     # 2D order: ρg, G, K
    assert r.iter == r_ref["iter"] == 21
    for k in ("norm_Rx", "norm_Ry", "norm_divV", "err_evo1"):
        assert np.allclose(getattr(r, k), r_ref[k], rtol=1e-10, atol=0), k
    d = ck.compare_stokes(env["down"](stokes), ref)
    assert max(d.values()) <= TOL_ITERS, d


def test_solcx_reference_test(env):
    """test/test_stokes_solcx.jl:26-37 : err_evo1[end] < 1e-8 at 32^2, Δη = 1e6"""
    jr, orc, ck = env["jr"], env["orc"], env["ck"]
    s = jr.miniapps.solcx2d(32)
    ref = _cp(s.arrays)
    r_ref = orc.stokes2d_solve(ref, ck.oracle_params2d(orc, s))
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    iters = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
    assert iters.err_evo1[-1] < 1.0e-8
    assert iters.iter == r_ref["iter"]
    dev = env["down"](stokes)
    for k in ("Vx", "Vy", "P"):
        assert ck.max_rel_diff(dev[k], ref[k]) < 1e-6, k


def test_elastic_buildup_reference_test(env):
    """test/test_stokes_elastic_buildup.jl:26-55 with finite dt and G: pins the τ_o / G·dt terms on the GPU.
    First 2 kyr (40 solves) of the reference's 10 kyr, step by step against the oracle (whose full 200-step run
    meets the reference's 5e-3 bound in tests/test_oracle_golden.py) and against the analytic build-up curve."""
    import math
    jr, orc, ck = env["jr"], env["orc"], env["ck"]
    s = jr.miniapps.elastic_buildup2d(32)
    kyr, η0, εbg, Gv = (s.extra[k] for k in ("kyr", "η0", "εbg", "G"))
    ref = _cp(s.arrays)
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    t = 0.0
    for step in range(40):
        dt = 0.05 * kyr
        r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, dt, None, kwargs=s.kwargs)
        s.dt = dt
        r_ref = orc.stokes2d_solve(ref, ck.oracle_params2d(orc, s))
        assert r.iter == r_ref["iter"], step
        t += dt
        got, want = float(stokes.τ.yy.abs().max()), float(np.abs(ref["tyy"]).max())
        assert got == pytest.approx(want, rel=1e-7), step
        sol = 2 * εbg * η0 * (1 - math.exp(-Gv * t / η0))
        assert abs(got - sol) / sol < 1e-2
    assert ck.max_rel_diff(jr.to_numpy(stokes.τ_o.yy), ref["toyy"]) < 1e-6


def _thermal_setup(jr, th, s):
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    for name, attr in (("T", "T"), ("Told", "Told"), ("H", "H"), ("qTx", "qTx"), ("qTy", "qTy"), ("qTx2", "qTx2"), ("qTy2", "qTy2"),
                       ("shear_heating", "shear_heating")):
        getattr(thermal, attr).copy_(from_numpy(s.arrays[name], dev))
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=s.pt["eps"])
    return thermal, pt, K, ρCp


def test_thermal_coefficients_and_bcs(env):
    jr, orc, th = env["jr"], env["orc"], env["th"]
    s = jr.miniapps.diffusion2d(32)
    thermal, pt, K, ρCp = _thermal_setup(jr, th, s)
    assert np.allclose(jr.to_numpy(pt.θr_dτ), s.arrays["thetar_dtau"], rtol=1e-14)
    assert np.allclose(jr.to_numpy(pt.dτ_ρ), s.arrays["dtau_rho"], rtol=1e-14)
    assert float(pt.θr_dτ[0, 0]) == pytest.approx(0.5156700149045073, rel=1e-12)
    assert float(pt.dτ_ρ[0, 0]) == pytest.approx(721404.061959508, rel=1e-12)
    b = s.flow_bcs
    p = orc.thermal_params2d(s.ni, s.grid._di["center"], s.dt, 1e-8, no_flux=b.no_flux, constant_value=b.constant_value,
                             constant_flux=b.constant_flux, periodic=b.periodic)
    Tref = s.arrays["T"].copy(order="F")
    orc.thermal_bcs2d(Tref, p)
    th.thermal_bcs_(thermal, b)
    assert np.array_equal(jr.to_numpy(thermal.T), Tref)


@pytest.mark.parametrize("form", ["array", "rheology"])
def test_thermal_iterations_match_oracle(env, form):
    jr, orc, th = env["jr"], env["orc"], env["th"]
    from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
    s = jr.miniapps.diffusion2d(24, iterMax=300, nout=100)
    b = s.flow_bcs
    rheo = s.extra["rheology"] if form == "rheology" else None
    p = orc.thermal_params2d(s.ni, s.grid._di["center"], s.dt, 1e-30, iterMax=300, nout=100, no_flux=b.no_flux,
                             constant_value=b.constant_value, constant_flux=b.constant_flux, periodic=b.periodic, rheology=rheo)
    orc.thermal_bcs2d(s.arrays["T"], p)
    add_perturbation(s.arrays["T"], s.grid, **s.extra["perturbation"])
    thermal, pt, K, ρCp = _thermal_setup(jr, th, s)
    pt.ϵ = 1e-30
    ref = _cp(s.arrays)
    r_ref = orc.heatdiffusion_PT2d(ref, p)
    A, B = (rheo, None) if form == "rheology" else (K, ρCp)
    r = jr.heatdiffusion_PT_(thermal, pt, b, A, B, s.dt, s.grid, kwargs=dict(iterMax=300, nout=100, verbose=False))
    assert list(r.iter_count) == list(r_ref["iter_count"]) == [100, 200, 300]
    assert np.allclose(r.norm_ResT, r_ref["norm_ResT"], rtol=1e-9)
    for name, t in (("T", thermal.T), ("Told", thermal.Told), ("dT", thermal.ΔT), ("qTx", thermal.qTx), ("qTy2", thermal.qTy2), ("ResT", thermal.ResT)):
        assert env["ck"].max_rel_diff(jr.to_numpy(t), ref[name]) <= TOL_ITERS, name


def test_diffusion2d_reference_test(env):
    """test/test_diffusion2D.jl:127-135 on the GPU: 20 steps of 50 kyr, T[18,18] / T[17,17] within atol 0.1"""
    jr, orc, th = env["jr"], env["orc"], env["th"]
    from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
    s = jr.miniapps.diffusion2d(32)
    b = s.flow_bcs
    p = orc.thermal_params2d(s.ni, s.grid._di["center"], s.dt, 1e-8, no_flux=b.no_flux, constant_value=b.constant_value,
                             constant_flux=b.constant_flux, periodic=b.periodic)
    orc.thermal_bcs2d(s.arrays["T"], p)            # host-side setup, as the reference's test does before the time loop
    add_perturbation(s.arrays["T"], s.grid, **s.extra["perturbation"])
    thermal, pt, K, ρCp = _thermal_setup(jr, th, s)
    for _ in range(s.extra["nt"]):
        jr.heatdiffusion_PT_(thermal, pt, b, s.extra["rheology"], None, s.dt, s.grid, kwargs=dict(verbose=False))
    T = jr.to_numpy(thermal.T)
    assert T[17, 17] == pytest.approx(1817.9448461176817, abs=1.0e-1)
    assert T[16, 16] == pytest.approx(1827.4674313638786, abs=1.0e-1)


def test_compute_dt_and_displacement_conversions_known_answers(env):
    """test/test_Utils.jl:145-149: ni = (4, 4), li = (1, 1), Vy = 10 everywhere: compute_dt(stokes, di, 0.1[, igg]) === 0.022500000000000003,
    compute_dt(stokes, di) ≈ the same; velocity2displacement! / displacement2velocity! (types/displacement.jl:2-60) are exact scalings"""
    jr, st = env["jr"], env["st"]
    stokes = jr.StokesArrays(jr.AMDGPUBackend, (4, 4))
    stokes.V.Vy.fill_(10.0)
    di = (0.25, 0.25)
    assert st.compute_dt_(stokes, di, 0.1) == 0.022500000000000003
    assert st.compute_dt_(stokes, di, 0.1, igg=object()) == 0.022500000000000003
    assert st.compute_dt_(stokes, di) == pytest.approx(0.022500000000000003)
    assert st.compute_dt_(stokes, di, 0.01) == 0.01
    rng = np.random.default_rng(0)
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    vx, vy = rng.standard_normal((5, 6)), rng.standard_normal((6, 5))
    stokes.V.Vx.copy_(from_numpy(vx, stokes.P.device)); stokes.V.Vy.copy_(from_numpy(vy, stokes.P.device))
    st.velocity2displacement_(stokes, 0.3)
    assert np.array_equal(jr.to_numpy(stokes.U.Ux), vx * 0.3) and np.array_equal(jr.to_numpy(stokes.U.Uy), vy * 0.3)
    stokes.V.Vx.zero_(); stokes.V.Vy.zero_()
    st.displacement2velocity_(stokes, 0.3, jr.VelocityBoundaryConditions())            # no-op for velocity BCs
    assert not jr.to_numpy(stokes.V.Vx).any()
    st.displacement2velocity_(stokes, 0.3, jr.DisplacementBoundaryConditions())
    assert np.array_equal(jr.to_numpy(stokes.V.Vx), (vx * 0.3) * (1.0 / 0.3))
    s3 = jr.StokesArrays(jr.AMDGPUBackend, (4, 5, 6))
    s3.V.Vz.fill_(-2.0); s3.V.Vx[2, 3, 1] = 8.0
    assert st.compute_dt_(s3, (0.5, 0.5, 0.1)) == min(0.5 * (1.0 / 8.0), 0.1 * (1.0 / 2.0)) * 0.9


@pytest.mark.parametrize("ni,bcs", [((130, 67), "free_slip"), ((64, 200), "no_slip"), ((200, 33), "none"), ((17, 9), "free_slip")])
def test_2d_kernel_variants_are_bit_identical(env, ni, bcs):
    """the fused one-launch 2D iteration (variant 3: velocity update + flow_bcs! by rule + stress update of the next iteration, ping-pong
    state, lazily applied flow_bcs!) against the two-kernel loop (variant 2): same operation order -> identical fields and residual
    history, on grids whose rows span several waves"""
    import ctypes as C
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields2d(ni, bcs=bcs, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    outs, its = [], []
    h = _lib.default_handle()
    try:
        for variant in (2, 3, 0):
            h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
            its.append((r.iter, tuple(r.err_evo1)))
            outs.append(env["down"](stokes))
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
    assert its[0] == its[1] == its[2] and its[0][0] == 24
    for v in (1, 2):
        for k in outs[0]:
            a, b = outs[0][k], outs[v][k]
            if k in ("Vx", "Vy", "Ux", "Uy"):      # the four ghost corners are not read by any stencil
                a, b = a.copy(), b.copy()
                for c in ((0, 0), (0, -1), (-1, 0), (-1, -1)):
                    a[c] = b[c]
            assert np.array_equal(a, b, equal_nan=True), (v, k)


def test_graph_replay_of_unobserved_iterations_changes_nothing(env):
    """option loop_graphs: runs of unobserved one-launch iterations of the 2D visco-elastic loop and of the 2D heat-diffusion loop replay as captured hipGraphs
    (32 iterations each); results, iteration counts and the fused-launch counters equal those of plain launches"""
    import ctypes as C
    jr, orc, th = env["jr"], env["orc"], env["th"]
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
    h = _lib.default_handle(0)

    def opt(k, v=None):
        if v is not None:
            h.call("jrx_set_option", C.c_char_p(k), C.c_int64(v))
        out = C.c_int64(0)
        h.call("jrx_get_option", C.c_char_p(k), C.byref(out))
        return out.value
    outs = []
    try:
        for g in (0, 1):
            opt(b"loop_graphs", g)
            s = jr.miniapps.solcx2d(64, iterMax=399, nout=200)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
            stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
            c0 = opt(b"stat_fused2d")
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
            fused = opt(b"stat_fused2d") - c0
            st = download_stokes(stokes)
            s2 = jr.miniapps.diffusion2d(64, iterMax=500, nout=250)
            b = s2.flow_bcs
            p = orc.thermal_params2d(s2.ni, s2.grid._di["center"], s2.dt, 1e-30, no_flux=b.no_flux, constant_value=b.constant_value, constant_flux=b.constant_flux,
                                     periodic=b.periodic)
            orc.thermal_bcs2d(s2.arrays["T"], p)
            add_perturbation(s2.arrays["T"], s2.grid, **s2.extra["perturbation"])
            thermal, pt, Kt, ρCp = _thermal_setup(jr, th, s2)
            pt.ϵ = 1e-30
            c1 = opt(b"stat_thermal_fused")
            rt = jr.heatdiffusion_PT_(thermal, pt, b, Kt, ρCp, s2.dt, s2.grid, kwargs=dict(iterMax=500, nout=250, verbose=False))
            outs.append((r.iter, list(r.err_evo1), fused, st, list(rt.norm_ResT), opt(b"stat_thermal_fused") - c1, jr.to_numpy(thermal.T), jr.to_numpy(thermal.qTx)))
    finally:
        opt(b"loop_graphs", 1)
    a, b_ = outs
    assert a[0] == b_[0] == 400 and a[1] == b_[1] and a[2] == b_[2] > 300
    for k in a[3]:
        assert np.array_equal(a[3][k], b_[3][k], equal_nan=True), k
    assert a[4] == b_[4] and a[5] == b_[5] > 400
    assert np.array_equal(a[6], b_[6]) and np.array_equal(a[7], b_[7])


@pytest.mark.parametrize("ni,bcs,dt", [((130, 67), "free_slip", 0.25), ((64, 200), "no_slip", 0.25), ((200, 33), "none", 0.25), ((17, 9), "free_slip", 0.25), ((63, 5), "no_slip", 0.25),
                                       ((130, 67), "free_slip", np.inf), ((64, 200), "no_slip", np.inf), ((200, 33), "none", np.inf), ((126, 9), "free_slip", np.inf), ((3, 3), "free_slip", np.inf)])
def test_2d_fused_iteration_with_batched_loads_is_bit_identical(env, ni, bcs, dt):
    """k_fused2d_b (every operand of the one-launch iteration requested up front, boundary cases as selects; dt = Inf: its viscous-limit instantiation, which does not load
    τ_o, P0, K, G, Q -- here random and non-zero) against the control-flow form k_fused2d and the two-kernel loop: identical fields and residual history, rows that span
    several waves, rows of exactly one and two wave segments, the smallest grid"""
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields2d(ni, bcs=bcs, dt=dt, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    outs, its, used = [], [], []
    h = _lib.default_handle()
    try:
        for variant, batch in ((2, 1), (3, 0), (3, 1)):
            h.set_option("kernel_variant", variant)
            h.set_option("fused2d_batch", batch)
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            c0 = (h.get_option("stat_fused2d"), h.get_option("stat_visc_checks"), h.get_option("stat_visc_fallbacks"))
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
            used.append((h.get_option("stat_fused2d") - c0[0], h.get_option("stat_visc_checks") - c0[1], h.get_option("stat_visc_fallbacks") - c0[2]))
            its.append((r.iter, tuple(r.err_evo1)))
            outs.append(env["down"](stokes))
    finally:
        h.set_option("kernel_variant", 0)
        h.set_option("fused2d_batch", 1)
    assert used[0][0] == 0 and used[1][0] > 0 and used[2][0] == used[1][0]
    assert used[2][1:] == ((1, 0) if np.isinf(dt) else (0, 0)) and used[1][1:] == (0, 0)       # the operand check runs once per solve of the viscous-limit form, and passes
    assert its[0] == its[1] == its[2] and its[0][0] == 24
    for v in (1, 2):
        for k in outs[0]:
            a, b = outs[0][k], outs[v][k]
            if k in ("Vx", "Vy", "Ux", "Uy"):      # the four ghost corners are not read by any stencil
                a, b = a.copy(), b.copy()
                for c in ((0, 0), (0, -1), (-1, 0), (-1, -1)):
                    a[c] = b[c]
            assert np.array_equal(a, b, equal_nan=True), (v, k)


@pytest.mark.parametrize("poison", ["toxx=nan", "toxy=inf", "P0=inf", "Q=nan", "K=0", "G=nan"])
def test_2d_viscous_limit_falls_back_when_an_unloaded_operand_is_not_harmless(env, poison):
    """as test_viscous_limit_falls_back_when_an_unloaded_operand_is_not_harmless in 3D: with dt = Inf a NaN / Inf in τ_o, P0, Q (or K, G = 0 or NaN) makes the reference's
    run NaN; the viscous-limit form of the one-launch 2D iteration never reads them, so the operand check sends the solve to the general form: same status and fields as the
    control-flow kernel, which loads everything"""
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields2d((70, 20), bcs="free_slip", dt=np.inf, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    name, val = poison.split("=")
    s.arrays[name][7, 5] = {"nan": np.nan, "inf": np.inf, "0": 0.0}[val]
    h = _lib.default_handle()
    res = []
    try:
        for batch in (1, 0):
            h.set_option("kernel_variant", 3)
            h.set_option("fused2d_batch", batch)
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            c0 = (h.get_option("stat_visc_checks"), h.get_option("stat_visc_fallbacks"))
            try:
                r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
                status = ("ok", r.iter)
            except _lib.JrxError as e:
                status = ("error", str(e))
            res.append((status, env["down"](stokes), (h.get_option("stat_visc_checks") - c0[0], h.get_option("stat_visc_fallbacks") - c0[1])))
    finally:
        h.set_option("kernel_variant", 0)
        h.set_option("fused2d_batch", 1)
    (st1, out1, d1), (st0, out0, d0) = res
    assert d1 == (1, 1) and d0 == (0, 0)
    assert st1 == st0, (st1, st0)            # (the 2D visco-elastic loop leaves with NaN norms where the 3D one raises: Stokes2D.jl has no error("NaN(s)"))
    assert np.isnan(out1["P"]).any()
    for k in ("P", "Vx", "Vy", "txx", "tyy", "txy"):
        a, b = out1[k].copy(), out0[k].copy()
        if k in ("Vx", "Vy"):
            for c in ((0, 0), (0, -1), (-1, 0), (-1, -1)):
                a[c] = b[c]
        assert np.array_equal(a, b, equal_nan=True), k
