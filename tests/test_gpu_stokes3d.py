"""GPU parity tests, 3D Stokes: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Tolerance: north_star asks for velocity/pressure/residual fields to match the reference CPU backend
"within a stated fp64 tolerance".  Stated here: 1e-12 relative to the field's max magnitude for a
single sweep (the kernels keep the reference's operation order and fma placement, so the observed
difference is usually 0), 1e-9 after tens of iterations, 1e-6 on converged solves (round-off
amplified by thousands of PT iterations of a stiff iteration).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_SWEEP = 1e-12
TOL_ITERS = 1e-9


def _cp(arrs):
    return {k: v.copy(order="F") for k, v in arrs.items()}


@pytest.fixture(scope="module")
def env(jr, oracle):
    import torch
    assert torch.cuda.is_available()
    from justrelax_jl_amd import checks, stokes
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    return dict(jr=jr, orc=oracle, checks=checks, st=stokes, up=upload_stokes, down=download_stokes)


# wide grids take the z-marching kernels (tile widths 64/128/256/512), narrow ones the per-node kernels
@pytest.mark.parametrize("ni", [(17, 19, 23), (32, 32, 32), (5, 4, 3), (64, 9, 7), (70, 21, 20), (130, 37, 19), (200, 10, 9), (400, 8, 11)])
def test_stress_sweep_matches_reference_kernels(env, ni):
    jr, orc, ck, st = env["jr"], env["orc"], env["checks"], env["st"]
    s = jr.miniapps.random_fields3d(ni)
    ref = _cp(s.arrays)
    p = ck.oracle_params3d(orc, s)
    import ctypes as C
    L = orc.lib()
    f = orc.fields3d(ref)
    L.orc_compute_divV3d(f.divV, f.Vx, f.Vy, f.Vz, *[C.c_int64(n) for n in ni], *[C.c_double(x) for x in s.grid._di["center"]])
    L.orc_compute_P3d(f.P, f.P0, f.RP, f.divV, f.Q, f.eta, f.K, f.G, C.c_int64(int(np.prod(ni))), C.c_double(s.dt),
                      C.c_double(s.pt.r), C.c_double(s.pt.θ_dτ))
    orc.call3d("orc_compute_strain_rate3d", ref, p)
    orc.call3d("orc_compute_tau3d", ref, p)
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    st.sweep_stress_(stokes, s.pt, s.grid, K, G, s.dt, diag=True)
    dev = env["down"](stokes)
    names = ["divV", "P", "RP", "exx", "eyy", "ezz", "eyz", "exz", "exy", "txx", "tyy", "tzz", "tyz", "txz", "txy"]
    d = ck.compare_stokes(dev, ref, names)
    assert max(d.values()) <= TOL_SWEEP, d
    # untouched inputs stay untouched
    for k in ("Vx", "Vy", "Vz", "toxx", "toxy", "eta"):
        assert np.array_equal(dev[k], s.arrays[k])


def test_stress_sweep_state_only_writes_only_state(env):
    jr, st = env["jr"], env["st"]
    s = jr.miniapps.random_fields3d((12, 11, 10))
    a, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    b, _, _, _ = env["up"](s, jr.AMDGPUBackend)
    st.sweep_stress_(a, s.pt, s.grid, K, G, s.dt, diag=True)
    st.sweep_stress_(b, s.pt, s.grid, K, G, s.dt, diag=False)
    da, db = env["down"](a), env["down"](b)
    for k in ("P", "txx", "tyy", "tzz", "tyz", "txz", "txy"):
        assert np.array_equal(da[k], db[k]), k
    for k in ("divV", "RP", "exx", "exy"):
        assert np.array_equal(db[k], s.arrays[k]), k      # diagnostics not written in state-only mode


@pytest.mark.parametrize("ni", [(17, 19, 23), (32, 32, 32), (3, 3, 3), (70, 21, 20), (130, 37, 19), (200, 10, 9), (400, 8, 11)])
def test_velocity_sweep_matches_compute_V(env, ni):
    jr, orc, ck, st = env["jr"], env["orc"], env["checks"], env["st"]
    s = jr.miniapps.random_fields3d(ni, seed=7)
    ref = _cp(s.arrays)
    p = ck.oracle_params3d(orc, s)
    etatau = orc.compute_maxloc(ref["eta"])
    import ctypes as C
    orc.call3d("orc_compute_V3d", ref, p, etatau.ctypes.data_as(C.POINTER(C.c_double)))
    orc.call3d("orc_velocity2displacement3d", ref, p)
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    et = jr.fzeros(ni, stokes.P.device)
    jr.compute_maxloc_(et, stokes.viscosity.η)
    assert np.array_equal(jr.to_numpy(et), etatau)
    st.sweep_velocity_(stokes, s.pt, s.grid, ρg, et, s.dt, diag=True)
    dev = env["down"](stokes)
    d = ck.compare_stokes(dev, ref, ["Vx", "Vy", "Vz", "Rx", "Ry", "Rz", "Ux", "Uy", "Uz"])
    assert max(d.values()) <= TOL_SWEEP, d


@pytest.mark.parametrize("kind", ["free_slip", "no_slip", "periodic", "mixed"])
def test_flow_bcs3d(env, kind):
    jr, orc = env["jr"], env["orc"]
    ni = (6, 7, 5)
    s = jr.miniapps.random_fields3d(ni, bcs=kind, seed=3)
    ref = _cp(s.arrays)
    b = s.flow_bcs
    orc.flow_bcs3d(ref["Vx"], ref["Vy"], ref["Vz"], ni, b.free_slip, b.no_slip, b.periodic)
    stokes, *_ = env["up"](s, jr.AMDGPUBackend)
    jr.flow_bcs_(stokes, b)
    dev = env["down"](stokes)
    for k in ("Vx", "Vy", "Vz"):
        assert np.array_equal(dev[k], ref[k]), k           # pure copies / negations: bit-exact, edges included
    if kind == "free_slip":                                  # the reference's own assertions (test_boundary_conditions3D.jl:88-99)
        Vx, Vy, Vz = dev["Vx"], dev["Vy"], dev["Vz"]
        assert np.array_equal(Vx[:, :, 0], Vx[:, :, 1]) and np.array_equal(Vx[:, :, -1], Vx[:, :, -2])
        assert np.array_equal(Vx[:, 0, :], Vx[:, 1, :]) and np.array_equal(Vx[:, -1, :], Vx[:, -2, :])
        assert np.array_equal(Vy[0], Vy[1]) and np.array_equal(Vy[-1], Vy[-2])
        assert np.array_equal(Vz[:, 0, :], Vz[:, 1, :]) and np.array_equal(Vz[0], Vz[1])


def test_residual_sumsq(env):
    jr, orc, ck, st = env["jr"], env["orc"], env["checks"], env["st"]
    s = jr.miniapps.random_fields3d((33, 20, 17), seed=11)
    p = ck.oracle_params3d(orc, s)
    want = orc.residual_sumsq3d(s.arrays, p)
    stokes, *_ = env["up"](s, jr.AMDGPUBackend)
    got = st.residual_sumsq(stokes, s.pt, s.grid)
    assert np.allclose(got, want, rtol=1e-13, atol=0)
    got2 = st.residual_sumsq(stokes, s.pt, s.grid)
    assert np.array_equal(got, got2)                       # deterministic reduction


@pytest.mark.parametrize("ni,bcs", [((17, 19, 23), "free_slip"), ((12, 12, 12), "no_slip"), ((10, 9, 8), "mixed"),
                                    ((16, 16, 16), "none"), ((100, 12, 10), "free_slip"), ((260, 9, 8), "no_slip")])
def test_solve_matches_oracle_over_iterations(env, ni, bcs):
    """Finite dt, G, K: every elastic/compressible term active.  21 iterations, checks every 5."""
    jr, orc, ck = env["jr"], env["orc"], env["checks"]
    s = jr.miniapps.random_fields3d(ni, bcs=bcs, iterMax=20, nout=5)
    s.pt.ϵ_rel, s.pt.ϵ_abs = 1e-30, 1e-30
    ref = _cp(s.arrays)
    r_ref = orc.stokes3d_solve(ref, ck.oracle_params3d(orc, s))
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 21
    assert np.array_equal(r.err_evo2, r_ref["err_evo2"])
    for k in ("norm_Rx", "norm_Ry", "norm_Rz", "norm_divV", "err_evo1"):
        assert np.allclose(getattr(r, k), r_ref[k], rtol=1e-10, atol=0), k
    dev = env["down"](stokes)
    d = ck.compare_stokes(dev, ref)
    assert max(d.values()) <= TOL_ITERS, d
    # final τ -> τ_o copy (multi_copy!, Stokes3D.jl:172-173)
    for c in ("xx", "yy", "zz", "yz", "xz", "xy"):
        assert np.array_equal(dev["to" + c], dev["t" + c])


def test_solvi3d_reference_test(env):
    """test/test_stokes_solvi3D.jl:25-55 : 16^3, iterMax=5000, nout=100 -> norm_Rx[end] < 1e-8."""
    jr, orc, ck = env["jr"], env["orc"], env["checks"]
    s = jr.miniapps.solvi3d(16)
    ref = _cp(s.arrays)
    r_ref = orc.stokes3d_solve(ref, ck.oracle_params3d(orc, s))
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    iters = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
    assert iters.norm_Rx[-1] < 1.0e-8
    assert iters.iter == r_ref["iter"] == 5001              # the loop always runs iterMax+1 iterations
    dev = env["down"](stokes)
    for k in ("Vx", "Vy", "Vz"):
        assert ck.max_rel_diff(dev[k], ref[k]) < 1e-6, k
    # ∇·V_bc = ε ≠ 0 with K = Inf: the mean pressure drifts every iteration; compare P - mean(P)
    assert ck.max_rel_diff(dev["P"] - dev["P"].mean(), ref["P"] - ref["P"].mean()) < 1e-6
    assert np.allclose(iters.norm_divV, r_ref["norm_divV"], rtol=1e-8)


def test_taylor_green_reference_test(env):
    """test/test_stokes_taylor_green.jl:29-41: PT err < 1e-8, order > 1.7, L2_v < 5e-3, L2_p < 1.5e-1."""
    jr = env["jr"]
    from justrelax_jl_amd.miniapps.stokes3d import taylor_green_error_norms
    errors = []
    for n in (8, 16):
        s = jr.miniapps.taylor_green3d(n)
        stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
        iters = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
        assert iters.err_evo1[-1] < 1.0e-8
        errors.append(taylor_green_error_norms(env["down"](stokes), s.grid))
    L2_p, L2_vx, L2_vy, L2_vz = errors[-1]
    order = np.log2(np.array(errors[0]) / np.array(errors[1]))
    assert (order > 1.7).all(), order
    assert max(L2_vx, L2_vy, L2_vz) < 5.0e-3 and L2_p < 1.5e-1


def test_burstedde_reference_test(env):
    """test/test_stokes_burstedde.jl:29-46 on the device (variable viscosity, body forces, velocity prescribed on all faces):
    PT err < 1e-8, velocity orders > 1.4, max L2_v < 3e-2, L2_p < 2e-1; and the fields agree with the oracle's converged ones."""
    jr, orc, ck = env["jr"], env["orc"], env["checks"]
    errors = []
    for n in (8, 16):
        s = jr.miniapps.burstedde3d(n)
        stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
        iters = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
        assert iters.err_evo1[-1] < 1.0e-8
        dev = env["down"](stokes)
        errors.append(jr.miniapps.burstedde_error_norms(dev, s.grid, s.extra["di"]))
        ref = _cp(s.arrays)
        r = orc.stokes3d_solve(ref, ck.oracle_params3d(orc, s))
        assert r["iter"] == iters.iter
        for k in ("Vx", "Vy", "Vz", "P", "txx", "txy"):
            assert ck.max_rel_diff(dev[k], ref[k]) <= 1e-9, k
    L2_p, L2_vx, L2_vy, L2_vz = errors[-1]
    order = np.log2(np.array(errors[0]) / np.array(errors[1]))
    assert (order[1:] > 1.4).all(), order
    assert max(L2_vx, L2_vy, L2_vz) < 3.0e-2 and L2_p < 2.0e-1


def test_nan_is_reported_like_the_reference(env):
    """error("NaN(s)") at a check iteration (Stokes3D.jl:162) -> JRX_ERR_NAN."""
    jr = env["jr"]
    from justrelax_jl_amd._lib import JrxError
    s = jr.miniapps.random_fields3d((8, 8, 8), iterMax=20, nout=5)
    s.arrays["txx"][3, 3, 3] = np.nan
    stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
    with pytest.raises(JrxError) as e:
        jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
    assert e.value.status == 1 and "NaN" in str(e.value)


def test_cpu_arrays_are_refused(jr):
    s = jr.miniapps.random_fields3d((6, 6, 6))
    from justrelax_jl_amd.miniapps.common import upload_stokes
    stokes, ρg, K, G = upload_stokes(s, jr.CPUBackend)
    with pytest.raises(NotImplementedError):
        jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)


@pytest.mark.parametrize("ni,bcs", [((130, 20, 17), "free_slip"), ((70, 12, 9), "free_slip"), ((130, 17, 20), "no_slip"),
                                    ((97, 9, 33), "none"), ((64, 16, 40), "no_slip"), ((200, 8, 8), "free_slip"),
                                    ((130, 18, 19), "slip_mix"), ((66, 9, 35), "slip_mix"), ((130, 12, 17), "periodic"), ((70, 20, 9), "mixed"),
                                    ((130, 9, 24), "mixed")])
def test_kernel_variants_are_bit_identical(env, ni, bcs):
    """auto (0), per-node kernels (1), the two z-marching sweeps (2) and the fused PT pipeline (3) must agree bit for bit:
    same operation order, and the fused kernel's on-the-fly low-face boundary rules reproduce flow_bcs!."""
    import ctypes as C
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d(ni, bcs=bcs, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    outs, its = [], []
    h = _lib.default_handle()
    try:
        # variant 13 / 3: the fused pipeline with the other tile shape (option fused_tile: 64 x 4 / 32 x 8 threads)
        tile0 = C.c_int64(0)
        h.call("jrx_tuning_get", C.c_char_p(b"fused_tile"), C.byref(tile0))
        for variant in (0, 1, 2, 3, 13):
            h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant % 10))
            h.call("jrx_tuning_set", C.c_char_p(b"fused_tile"), C.c_int64((0 if 62 < ni[0] <= 90 else 1) if variant >= 10 else tile0.value))      # the shape the default rule does not pick for this nx
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            n0, n1 = C.c_int64(0), C.c_int64(0)
            h.call("jrx_get_option", C.c_char_p(b"stat_fused3d"), C.byref(n0))
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
            h.call("jrx_get_option", C.c_char_p(b"stat_fused3d"), C.byref(n1))
            if variant % 10 == 3:
                assert n1.value > n0.value, "the fused pipeline did not run (periodic faces included)"
            its.append((r.iter, tuple(r.err_evo1)))
            outs.append(env["down"](stokes))
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
        h.call("jrx_tuning_set", C.c_char_p(b"fused_tile"), tile0)
    assert its[0] == its[1] == its[2] == its[3] == its[4] and its[0][0] == 24
    for v in (1, 2, 3, 4):
        for k in outs[0]:
            m = env["checks"].interior_mask3d(k, outs[0][k].shape)
            assert np.array_equal(outs[0][k][m], outs[v][k][m], equal_nan=True), (v, k)


@pytest.mark.parametrize("ni,bcs,tile", [((130, 20, 17), "free_slip", 0), ((97, 9, 33), "none", 0), ((130, 17, 20), "no_slip", 1), ((66, 9, 35), "slip_mix", 1),
                                         ((130, 12, 17), "periodic", 0), ((130, 20, 17), "free_slip", 3), ((97, 23, 33), "none", 3), ((130, 17, 20), "no_slip", 3),
                                         ((70, 30, 19), "slip_mix", 3), ((130, 12, 17), "periodic", 3), ((190, 50, 9), "slip_mix", 3),
                                         ((130, 40, 17), "free_slip", 4), ((97, 23, 33), "none", 4), ((130, 17, 20), "no_slip", 4), ((70, 50, 19), "slip_mix", 4), ((130, 12, 17), "periodic", 4)])
def test_viscous_limit_kernel_equals_the_general_one(env, ni, bcs, tile):
    """dt = Inf (SolVi3D, Burstedde, TaylorGreen: SolVi3D.jl:96 hands dt = Inf): the fused kernel's viscous-limit form does not load τ_o, P0, K, G, Q
    -- every one of them random and non-zero here -- and must equal the general fused kernel and the per-node kernels, which do."""
    import ctypes as C
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d(ni, bcs=bcs, dt=np.inf, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    h = _lib.default_handle()
    tile0 = C.c_int64(0)
    h.call("jrx_tuning_get", C.c_char_p(b"fused_tile"), C.byref(tile0))
    outs, its = [], []
    try:
        h.call("jrx_tuning_set", C.c_char_p(b"fused_tile"), C.c_int64(tile))
        for variant, visc in ((3, 1), (3, 0), (2, 1), (2, 0), (1, 1)):      # 2: the z-marching stress sweep has the same form
            h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
            h.call("jrx_set_option", C.c_char_p(b"viscous_limit"), C.c_int64(visc))
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            n0, n1 = C.c_int64(0), C.c_int64(0)
            h.call("jrx_get_option", C.c_char_p(b"stat_fused3d"), C.byref(n0))
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
            h.call("jrx_get_option", C.c_char_p(b"stat_fused3d"), C.byref(n1))
            assert (n1.value > n0.value) == (variant == 3)
            its.append((r.iter, tuple(r.err_evo1)))
            outs.append(env["down"](stokes))
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
        h.call("jrx_set_option", C.c_char_p(b"viscous_limit"), C.c_int64(1))
        h.call("jrx_tuning_set", C.c_char_p(b"fused_tile"), tile0)
    assert its[0] == its[1] == its[2] == its[3] == its[4] and its[0][0] == 24
    for v in (1, 2, 3, 4):
        for k in outs[0]:
            m = env["checks"].interior_mask3d(k, outs[0][k].shape)
            assert (k[0] == "U" or np.isfinite(outs[0][k][m]).all()) and np.array_equal(outs[0][k][m], outs[v][k][m], equal_nan=True), (v, k)      # U = V dt = ±Inf, NaN where V = 0


@pytest.mark.parametrize("ym", [2, 4])
@pytest.mark.parametrize("ni,bcs,zero", [((130, 40, 17), "free_slip", "xyz"), ((97, 23, 33), "none", ""), ((70, 50, 19), "slip_mix", "xy"), ((190, 57, 9), "no_slip", "xyz"), ((130, 12, 17), "periodic", "")])
def test_y_march_of_the_fused_kernel_keeps_the_bits(env, ni, bcs, zero, ym):
    """round 6: a block of the one-launch viscous-limit kernel (64 x 8 tile) marches `fused_ym` tile rows in y; row 0 of every tile behind the first takes the new velocities and
    η of its cells from LDS, where the top row of the previous tile left them, instead of recomputing them from re-loaded operands (the y halo).  Same values: every array equals
    the one-tile-per-block form and the per-node kernels bit for bit -- for ny that fill whole marches, leave a shorter last march (ny = 23: 4 tile rows; 57: 9) or a single tile
    row (ny = 12: no march at all)."""
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d(ni, bcs=bcs, dt=np.inf, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    for c in zero:
        s.arrays["f" + c][...] = 0.0
    h = _lib.default_handle()
    tile0, ym0 = h.get_option("fused_tile"), h.get_option("fused_ym")
    outs, its, marched = [], [], []
    try:
        h.set_option("fused_tile", 3)
        for variant, m in ((3, ym), (3, 0), (1, 0)):
            h.set_option("kernel_variant", variant)
            h.set_option("fused_ym", m)
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            c0 = h.get_option("stat_fused3d_ym")
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
            marched.append(h.get_option("stat_fused3d_ym") - c0)
            its.append((r.iter, tuple(r.err_evo1)))
            outs.append(env["down"](stokes))
    finally:
        h.set_option("kernel_variant", 0)
        h.set_option("fused_tile", tile0)
        h.set_option("fused_ym", ym0)
    assert (marched[0] > 0) == (bcs != "periodic") and marched[1:] == [0, 0], marched       # periodic faces: the kernel is not the one-launch form
    assert its[0] == its[1] == its[2] and its[0][0] == 24
    for v in (1, 2):
        for k in outs[0]:
            m = env["checks"].interior_mask3d(k, outs[0][k].shape)
            assert np.array_equal(outs[0][k][m], outs[v][k][m], equal_nan=True), (v, k)


@pytest.mark.parametrize("zero,nof", [("xy", 1), ("xyz", 2), ("xy one entry -0.0", 0), ("xy one entry 1e-300", 0), ("z", 0), ("xz", 0), ("", 0)])
@pytest.mark.parametrize("ni,bcs,tile,dt", [((130, 20, 17), "free_slip", 0, np.inf), ((66, 9, 35), "slip_mix", 1, np.inf), ((97, 9, 33), "none", 0, np.inf),
                                            ((130, 20, 17), "free_slip", 0, 0.25), ((66, 9, 35), "slip_mix", 1, 0.25), ((130, 17, 20), "no_slip", 0, 0.25),
                                            ((130, 20, 17), "free_slip", 3, np.inf), ((97, 23, 20), "none", 3, np.inf), ((130, 20, 17), "free_slip", 3, 0.25), ((70, 30, 19), "slip_mix", 3, 0.25),
                                            ((130, 40, 17), "free_slip", 4, np.inf), ((70, 50, 19), "slip_mix", 4, 0.25)])
def test_body_forces_that_are_zero_are_not_loaded_and_the_bits_stay(env, ni, bcs, tile, dt, zero, nof):
    """SolVi3D.jl:102 hands three ρg arrays of zeros, and every 3D model of the reference has ρg_x = ρg_y = 0 (gravity along z).  The one-launch viscous-limit kernel does not
    load body-force arrays in which the operand pass of the driver call has found nothing but +0.0 (all 64 bits zero): x - 0.5 (0 + 0) = x for every x, -0.0 and NaN included.
    A single -0.0 (x - (-0.0) turns x = -0.0 into +0.0) or a denormal keeps the loads.  Every case equals the per-node general kernels and the same form with the loads."""
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d(ni, bcs=bcs, dt=dt, iterMax=23, nout=7)      # finite dt: the general form of the fused kernel has the same instantiations
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    comps = zero.split(" ")[0]
    for c in comps:
        s.arrays["f" + c][...] = 0.0
    if "-0.0" in zero:
        s.arrays["fy"][ni[0] // 2, ni[1] - 1, 3] = -0.0
    if "1e-300" in zero:
        s.arrays["fx"][0, 0, 0] = 1e-300
    # velocities / stresses with zeros of both signs in them, so that a dropped "- 0.0" would show
    s.arrays["Vx"][3:9, 2:6, 2:9] = 0.0
    s.arrays["txx"][2:9, 2:6, 2:9] = -0.0
    h = _lib.default_handle()
    tile0 = h.get_option("fused_tile")
    outs, its, cnt = [], [], []
    try:
        h.set_option("fused_tile", tile)
        for variant, zf in ((3, 1), (3, 0), (1, 1)):
            h.set_option("kernel_variant", variant)
            h.set_option("zero_forces", zf)
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            c0 = [h.get_option(k) for k in ("stat_fused3d", "stat_fused3d_nof1", "stat_fused3d_nof2")]
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
            c1 = [h.get_option(k) for k in ("stat_fused3d", "stat_fused3d_nof1", "stat_fused3d_nof2")]
            cnt.append([b - a for a, b in zip(c0, c1)])
            its.append((r.iter, tuple(r.err_evo1)))
            outs.append(env["down"](stokes))
    finally:
        h.set_option("kernel_variant", 0)
        h.set_option("zero_forces", 1)
        h.set_option("fused_tile", tile0)
    assert cnt[0][0] > 0 and cnt[0][0] == cnt[1][0] and cnt[1][1:] == [0, 0] and cnt[2] == [0, 0, 0], cnt
    assert cnt[0][1] == (cnt[0][0] if nof == 1 else 0) and cnt[0][2] == (cnt[0][0] if nof == 2 else 0), (cnt, nof)
    assert its[0] == its[1] == its[2] and its[0][0] == 24
    for v in (1, 2):
        for k in outs[0]:
            m = env["checks"].interior_mask3d(k, outs[0][k].shape)
            # array_equal treats -0.0 == +0.0: compare the bit patterns
            a, b = np.ascontiguousarray(outs[0][k][m]), np.ascontiguousarray(outs[v][k][m])
            if k[0] == "U":      # U = V dt = ±Inf, NaN where V = 0
                assert np.array_equal(a, b, equal_nan=True), (v, k)
            elif v == 1 and k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "Rx", "Ry", "Rz", "RP"):
                # the same kernels with the loads: every bit of the state and the residuals, the signs of zeros included
                assert np.isfinite(a).all() and np.array_equal(a.view(np.uint64), b.view(np.uint64)), (v, k)
            else:
                # the per-node kernels, and the strain rates of observed iterations: exact zeros on boundary nodes carry the sign the ghost entries on the block's edges happen to
                # have (no-slip negates them; the one-launch flow_bcs! of unobserved iterations writes those entries from two faces), so these compare as numbers
                assert np.isfinite(a).all() and np.array_equal(a, b), (v, k)


@pytest.mark.parametrize("name,where,val", [("P0", "first", np.nan), ("P0", "last", np.inf), ("toxy", "last", np.nan), ("toyz", "first", np.inf), ("G", "last", 0.0), ("eta", "last", np.inf),
                                            ("fz", "last", 1e-300), ("fx", "first", -0.0)])
def test_operand_pass_sees_the_first_and_the_last_entry_of_every_array(env, name, where, val):
    """round 6: the operand pass streams every array on its own in 16-byte pairs; the entries outside the pairs -- the last one of an array of odd length (65 x 9 x 9 = 5,265
    cells, 66 x 9 x 10 nodes ...) -- and the very first one are looked at all the same: a poisoned one sends the call to the general kernels, a body-force entry that is not
    +0.0 keeps the loads"""
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d((65, 9, 9), bcs="free_slip", dt=np.inf, iterMax=9, nout=4)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    for c in "xyz":
        s.arrays["f" + c][...] = 0.0
    flat = s.arrays[name].reshape(-1, order="F")
    assert flat.base is s.arrays[name] or flat.base is s.arrays[name].base
    flat[0 if where == "first" else -1] = val
    h = _lib.default_handle()
    try:
        h.set_option("kernel_variant", 3)
        stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
        keys = ("stat_visc_checks", "stat_visc_fallbacks", "stat_fused3d", "stat_fused3d_nof1", "stat_fused3d_nof2")
        c0 = [h.get_option(k) for k in keys]
        try:
            jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
        except _lib.JrxError as e:
            assert "NaN" in str(e)
        d = [b - a for a, b in zip(c0, [h.get_option(k) for k in keys])]
    finally:
        h.set_option("kernel_variant", 0)
    assert d[0] == 1 and d[2] > 0
    if name in ("fx", "fz"):
        assert d[1] == 0 and d[4] == 0 and d[3] == (d[2] if name == "fz" else 0), d        # fz not zero: the x, y loads are still dropped (NOF = 1); fx not +0.0: every load stays
    else:
        assert d[1] == 1 and d[3] == d[4] == 0, d                                               # the check failed: general kernels


@pytest.mark.parametrize("poison", ["toxx=nan", "toyz=inf", "P0=inf", "Q=nan", "K=0", "G=nan", "none"])
def test_viscous_limit_falls_back_when_an_unloaded_operand_is_not_harmless(env, poison):
    """VERDICT r3 P3.  With dt = Inf the reference still multiplies τ_o, P0, Q by 0 and divides by K dt, G dt: a NaN / Inf in one of them (or K, G = 0: 0 * Inf)
    makes the residuals NaN and the driver raises error("NaN(s)") (Stokes3D.jl:162).  The viscous-limit kernels never read those arrays, so the library checks
    them once per driver call and runs the general kernels when an entry is not harmless: same status, same fields as with option viscous_limit = 0, and the
    oracle agrees that the run is NaN.  `none`: the check passes and the viscous form runs."""
    import ctypes as C
    jr, orc = env["jr"], env["orc"]
    from justrelax_jl_amd import _lib, checks
    s = jr.miniapps.random_fields3d((130, 20, 17), bcs="free_slip", dt=np.inf, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    if poison != "none":
        name, val = poison.split("=")
        s.arrays[name][7, 5, 3] = {"nan": np.nan, "inf": np.inf, "0": 0.0}[val]
    h = _lib.default_handle()
    res = []
    try:
        for visc in (1, 0):
            h.set_option("kernel_variant", 3)
            h.set_option("viscous_limit", visc)
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            c0 = [h.get_option(k) for k in ("stat_fused3d_visc", "stat_visc_checks", "stat_visc_fallbacks", "stat_fused3d")]
            try:
                r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
                status = ("ok", r.iter)
            except _lib.JrxError as e:
                status = ("error", str(e))
            c1 = [h.get_option(k) for k in ("stat_fused3d_visc", "stat_visc_checks", "stat_visc_fallbacks", "stat_fused3d")]
            res.append((status, env["down"](stokes), [b - a for a, b in zip(c0, c1)]))
    finally:
        h.set_option("kernel_variant", 0)
        h.set_option("viscous_limit", 1)
    (st1, out1, d1), (st0, out0, d0) = res
    assert d0[:3] == [0, 0, 0] and d0[3] > 0                     # option off: no check, general form of the fused kernel
    assert d1[1] == 1 and d1[3] > 0
    if poison == "none":
        assert st1[0] == st0[0] == "ok" and d1[0] == d1[3] and d1[2] == 0            # every fused launch was the viscous form
    else:
        assert d1[0] == 0 and d1[2] == 1                                                # the check failed: not one launch of the viscous form
        assert st1[0] == st0[0] == "error" and "NaN" in st1[1] and "JRX_ERR_NAN" in st1[1], (st1, st0)
        ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
        r_ref = orc.stokes3d_solve(ref, checks.oracle_params3d(orc, s))
        assert np.isnan(r_ref["err_evo1"][-1])                                          # the restated reference run is NaN, too
    for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy"):
        m = env["checks"].interior_mask3d(k, out1[k].shape)
        assert np.array_equal(out1[k][m], out0[k][m], equal_nan=True), k


@pytest.mark.parametrize("gh", [0, 3, 4])
@pytest.mark.parametrize("ni,bcs,zero", [((130, 20, 17), "free_slip", ""), ((97, 9, 33), "none", "xy"), ((130, 17, 20), "no_slip", "xyz"), ((66, 12, 35), "slip_mix", ""), ((190, 50, 9), "slip_mix", "xy")])
def test_general_form_folds_its_high_face_layers(env, ni, bcs, zero, gh):
    """for any dt the fused kernel now updates the stress nodes on the high faces i = nx, j = ny, k = nz itself (one launch per unobserved iteration instead of the
    kernel + the boundary-layer launch), built for four or three waves per SIMD (tuning switch general_hif; 0 = the launch pair of rounds 1-4).  Every form equals the per-node kernels."""
    jr = env["jr"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d(ni, bcs=bcs, dt=0.25, iterMax=23, nout=7)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    for c in zero:
        s.arrays["f" + c][...] = 0.0
    h = _lib.default_handle()
    outs, its, cnt = [], [], []
    try:
        h.set_option("general_hif", gh)
        for variant in (3, 1):
            h.set_option("kernel_variant", variant)
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            c0 = [h.get_option(k) for k in ("stat_fused3d", "stat_fused3d_general_hif")]
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
            cnt.append([h.get_option(k) - c for k, c in zip(("stat_fused3d", "stat_fused3d_general_hif"), c0)])
            its.append((r.iter, tuple(r.err_evo1)))
            outs.append(env["down"](stokes))
    finally:
        h.set_option("kernel_variant", 0)
        h.set_option("general_hif", 0)
    one_launch = gh != 0 and ni[0] > 90          # the one-launch instantiations exist for the 64 x 4 tile (nx = 63 .. 90 runs 32 x 8 tiles)
    assert cnt[0][0] > 0 and cnt[0][1] == (cnt[0][0] if one_launch else 0) and cnt[1] == [0, 0], cnt
    assert its[0] == its[1] and its[0][0] == 24
    for k in outs[0]:
        m = env["checks"].interior_mask3d(k, outs[0][k].shape)
        assert np.array_equal(outs[0][k][m], outs[1][k][m], equal_nan=True), (gh, k)


def test_operand_cache_reuses_the_verdict_until_the_fields_are_declared_dirty(env):
    """option operand_cache = 1: the operand pass of the 3D visco-elastic drivers runs once per (operand pointers, extents, dt); the next driver call on the
    same arrays reuses its verdict -- same bits as a call that looks again -- until jrx_fields_dirty, after which a poisoned operand is found and the general kernels report the
    reference's NaN.  solve! itself invalidates the verdict (it writes τ_o at its end)."""
    jr, st = env["jr"], env["st"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d((130, 20, 17), bcs="free_slip", dt=np.inf, iterMax=23, nout=7)
    for c in "xyz":
        s.arrays["f" + c][...] = 0.0
    h = _lib.default_handle()
    keys = ("stat_visc_checks", "stat_operand_cache_hits", "stat_fused3d_visc", "stat_fused3d_nof2", "stat_fused3d")
    cnt = lambda: [h.get_option(k) for k in keys]
    outs = []
    try:
        for cache in (0, 1):
            h.set_option("operand_cache", cache)
            h.set_option("kernel_variant", 3)
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            ητ = jr.fzeros(s.ni, stokes.P.device)
            jr.compute_maxloc_(ητ, stokes.viscosity.η, handle=h)
            c0 = cnt()
            for _ in range(3):
                st.iterate_timed_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, ητ, s.dt, 6, handle=h)
            d = [b - a for a, b in zip(c0, cnt())]
            assert d[0] == (1 if cache else 3) and d[1] == (2 if cache else 0), d
            assert d[2] == d[4] > 0 and d[3] == d[4], d                   # every fused launch in the viscous-limit form without body-force loads, cached verdict or not
            outs.append(env["down"](stokes))
            if cache:
                # a NaN written behind the library's back stays unseen (that is the contract) ...
                K[5, 5, 5] = float("nan")
                c0 = cnt()
                st.iterate_timed_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, ητ, s.dt, 6, handle=h)
                d = [b - a for a, b in zip(c0, cnt())]
                assert d[0] == 0 and d[1] == 1 and d[2] == d[4] > 0, d
                # ... until the caller says so: the pass runs again, fails, and the general kernels run
                h.call("jrx_fields_dirty")
                c0 = cnt()
                st.iterate_timed_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, ητ, s.dt, 6, handle=h)
                d = [b - a for a, b in zip(c0, cnt())]
                assert d[0] == 1 and d[1] == 0 and d[2] == 0 and d[4] > 0, d
                # solve! writes τ_o at its end: the verdict it left is not reused (fresh arrays: the poisoned run above has left NaNs in the state)
                stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
                s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
                c0 = cnt()
                jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs, handle=h)
                jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs, handle=h)
                d = [b - a for a, b in zip(c0, cnt())]
                assert d[0] == 2 and d[1] == 0, d
    finally:
        h.set_option("operand_cache", 0)
        h.set_option("kernel_variant", 0)
    for k in outs[0]:
        m = env["checks"].interior_mask3d(k, outs[0][k].shape)
        assert np.array_equal(outs[0][k][m], outs[1][k][m], equal_nan=True), k


def test_iterate_timed_leaves_state_in_user_arrays(env):
    """bench hook: K back-to-back iterations through the fused pipeline == K iterations of the per-node kernels."""
    import ctypes as C
    jr, st = env["jr"], env["st"]
    from justrelax_jl_amd import _lib
    s = jr.miniapps.random_fields3d((130, 16, 12), iterMax=5, nout=100)
    h = _lib.default_handle()
    outs = []
    try:
        for variant in (3, 1, 2):
            h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
            stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
            et = jr.fzeros(s.ni, stokes.P.device)
            jr.compute_maxloc_(et, stokes.viscosity.η)
            t = st.iterate_timed_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, et, s.dt, 9)
            assert t[0] > 0 and (t[3] > 0) == (variant == 3) and (t[4] > 0) == (variant == 3) and t[4] <= t[3]
            outs.append(env["down"](stokes))
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
    for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy"):
        m = env["checks"].interior_mask3d(k, outs[0][k].shape)
        assert np.array_equal(outs[0][k][m], outs[1][k][m]) and np.array_equal(outs[0][k][m], outs[2][k][m]), k


@pytest.mark.parametrize("dt", [0.25, np.inf])
@pytest.mark.parametrize("ni,bcs,iters", [((130, 16, 12), "free_slip", 10), ((130, 16, 12), "none", 2), ((66, 9, 35), "slip_mix", 4), ((97, 9, 33), "no_slip", 20), ((130, 16, 12), "free_slip", 9)])
def test_iterate_timed_with_an_odd_number_of_fused_steps_ends_in_the_user_arrays_without_a_copy(env, ni, bcs, iters, dt):
    """A batch of K iterations is the first stress sweep, K - 1 fused steps and the last velocity sweep; every fused step moves the state to the other set.  With K - 1 odd the
    two end sweeps write out of place (P and the stresses into the scratch set first, the velocities back last) so that the batch still ends in the caller's arrays -- no copy-back,
    no extra un-fused iteration (tuning switch end_flips).  Same fields as with the switch off and as K iterations of the per-node kernels, for two different problems in a row on
    one handle (the scratch set still holds the previous problem's state: an entry the out-of-place sweeps did not write would show)."""
    jr, st = env["jr"], env["st"]
    from justrelax_jl_amd import _lib
    h = _lib.default_handle()
    try:
        for seed in (11, 12):
            s = jr.miniapps.random_fields3d(ni, seed=seed, bcs=bcs, dt=dt, iterMax=5, nout=100)
            if seed == 12:
                s.arrays["fx"][...] = 0.0; s.arrays["fy"][...] = 0.0
            outs, fused = [], []
            for variant, flips in ((3, 1), (3, 0), (1, 1)):
                h.set_option("kernel_variant", variant)
                h.set_option("end_flips", flips)
                stokes, ρg, K, G = env["up"](s, jr.AMDGPUBackend)
                et = jr.fzeros(s.ni, stokes.P.device)
                jr.compute_maxloc_(et, stokes.viscosity.η)
                c0 = h.get_option("stat_fused3d")
                st.iterate_timed_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, et, s.dt, iters)
                fused.append(h.get_option("stat_fused3d") - c0)
                outs.append(env["down"](stokes))
            odd = (iters - 1) % 2 == 1
            assert fused == [iters - 1, iters - 2 if odd else iters - 1, 0], fused
            for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy"):
                m = env["checks"].interior_mask3d(k, outs[0][k].shape)
                assert np.isfinite(outs[0][k][m]).all() and np.array_equal(outs[0][k][m], outs[1][k][m]) and np.array_equal(outs[0][k][m], outs[2][k][m]), (seed, k)
    finally:
        h.set_option("kernel_variant", 0)
        h.set_option("end_flips", 1)


@pytest.mark.parametrize("axis,steps", [(2, 40), (1, 20), (0, 20)])
def test_3d_elastic_buildup_on_the_device(jr, oracle, axis, steps):
    """the τ_o / 1/(G dt) terms of the 3D kernels on the device, anchored on the reference's elastic build-up (Elastic_BuildUp.jl:4,55-56,75-86;
    test_stokes_elastic_buildup.jl:47-54) through its plane-strain restatement (tests/test_oracle_plane_strain3d.py pins the 3D oracle on the reference's
    5e-3 bound and on the 2D run) in all three orientations: uniform along z (τxx, τyy, τxy), along y (τxx, τzz, τxz) and along x (τyy, τzz, τyz).  The first
    solves of 1000 iterations, step by step against the 3D oracle, the pinned 2D oracle and the analytic curve."""
    import math
    from justrelax_jl_amd import checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    orc = oracle
    s2 = jr.miniapps.elastic_buildup2d(32)
    s3 = jr.miniapps.plane_strain3d(s2, nz=3, axis=axis)
    nm = s3.extra["names"]
    kyr, η0, εbg, Gv = (s2.extra[k] for k in ("kyr", "η0", "εbg", "G"))
    ref3 = {k: v.copy(order="F") for k, v in s3.arrays.items()}
    stokes, ρg, K, G = upload_stokes(s3, jr.AMDGPUBackend)
    t = 0.0
    for step in range(steps):
        dt = 0.05 * kyr
        s2.dt = s3.dt = dt
        r = jr.solve_(stokes, s3.pt, s3.grid, s3.flow_bcs, ρg, K, G, dt, None, kwargs=s3.kwargs)
        r3 = orc.stokes3d_solve(ref3, checks.oracle_params3d(orc, s3))
        orc.stokes2d_solve(s2.arrays, checks.oracle_params2d(orc, s2))
        assert r.iter == r3["iter"] == 1000, step
        t += dt
        got = float(getattr(stokes.τ, nm["tyy"][1:]).abs().max())
        assert got == pytest.approx(float(np.abs(ref3[nm["tyy"]]).max()), rel=1e-9), step
        assert got == pytest.approx(float(np.abs(s2.arrays["tyy"]).max()), rel=1e-9), step
        sol = 2 * εbg * η0 * (1 - math.exp(-Gv * t / η0))
        assert abs(got - sol) / sol < 2e-2
    out = download_stokes(stokes)
    for k in ("txx", "tyy", "txy", "toxx", "toyy", "toxy", "P", "Vx", "Vy"):
        scale = np.abs(s2.arrays[k]).max()
        assert np.abs(out[nm[k]] - np.expand_dims(s2.arrays[k], axis)).max() <= 1e-9 * max(scale, 1e-300), k
        assert checks.max_rel_diff(out[nm[k]], ref3[nm[k]]) <= 1e-9, k
    assert np.abs(out[nm["toyy"]]).max() > (0.15 if steps == 40 else 0.07) * 2 * εbg * η0


# (variant, 2D grid, cells along the uniform axis): 1 = one node per thread; 3 = the fused iteration kernel (its z-march carries the previous planes of V, η, G
# and the k - 1 shear stresses in registers / LDS); 2 on a block large enough for the z-marching sweeps (> 681,472 cells, launch_stress)
_PS_CASES = [(1, (70, 14), 9), (3, (70, 14), 9), (2, (96, 80), 96)]


@pytest.mark.parametrize("axis", [2, 1, 0])
@pytest.mark.parametrize("variant,n2,nu", _PS_CASES)
def test_3d_compressible_iterations_on_uniform_fields_equal_the_2d_oracle(jr, oracle, variant, n2, nu, axis):
    """compute_P!'s 1/(K dt) term and the τ_o terms of the stresses, finite K, G, dt, random fields uniform along one axis: 30 device iterations of the 3D
    path == the pinned 2D oracle on every plane (see tests/test_oracle_plane_strain3d.py).  Uniform along z anchors τxx, τyy, τxy; along y τxx, τzz, τxz; along x
    τyy, τzz, τyz -- the last two put a 2D stress on the components the z-marching kernels carry from plane to plane."""
    import ctypes as C
    from justrelax_jl_amd import _lib, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    orc = oracle
    if axis == 0 and variant == 3:
        n2, nu = (20, 14), 50                  # the fused kernel wants nx >= 48: the uniform axis is x here
    s2 = jr.miniapps.random_fields2d(n2, seed=77, iterMax=29, nout=10)
    s2.pt.ϵ_rel = s2.pt.ϵ_abs = 1e-30
    s2.arrays["eta"][...] = 0.37
    orc.flow_bcs2d(s2.arrays["Vx"], s2.arrays["Vy"], s2.ni, **{k: getattr(s2.flow_bcs, k) for k in ("free_slip", "no_slip", "periodic")})
    s3 = jr.miniapps.plane_strain3d(s2, nz=nu, axis=axis)
    nm = s3.extra["names"]
    h = _lib.default_handle()
    h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
    fused0 = h.get_option("stat_fused3d")
    try:
        stokes, ρg, K, G = upload_stokes(s3, jr.AMDGPUBackend)
        r = jr.solve_(stokes, s3.pt, s3.grid, s3.flow_bcs, ρg, K, G, s3.dt, None, kwargs=s3.kwargs)
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
    assert (h.get_option("stat_fused3d") > fused0) == (variant == 3)
    r2 = orc.stokes2d_solve(s2.arrays, checks.oracle_params2d(orc, s2))
    assert r.iter == r2["iter"] == 30
    out = download_stokes(stokes)
    for k in ("P", "Vx", "Vy", "txx", "tyy", "txy", "toxx", "toyy", "toxy", "exx", "eyy", "exy", "RP", "Rx", "Ry", "divV"):
        scale = max(np.abs(s2.arrays[k]).max(), 1e-300)
        assert np.abs(out[nm[k]] - np.expand_dims(s2.arrays[k], axis)).max() <= 1e-11 * scale, k
    oop = s3.extra["out_of_plane"]
    assert np.abs(out["t" + oop["tn"]]).max() > 0.0 and np.abs(out[oop["V"]]).max() == 0.0
    assert all(np.abs(out["t" + c]).max() == 0.0 for c in oop["shear"])


@pytest.mark.parametrize("ni,dt,bcs", [((130, 96, 100), 0.25, "free_slip"), ((130, 96, 100), float("inf"), "free_slip"), ((70, 40, 36), 0.25, "no_slip"), ((200, 64, 72), float("inf"), "free_slip")])
def test_the_second_state_set_may_hold_anything(jr, ni, dt, bcs):
    """the fused pipeline ping-pongs between the caller's arrays and a set of the library's own: whatever that set holds when it is allocated (here: NaNs in every entry, test switch
    "scratch_poison") must not reach a result -- every entry a kernel reads has been written by a kernel before"""
    from justrelax_jl_amd import _lib, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    outs = []
    for poison in (0, 1):
        h = _lib.Handle(0)
        try:
            h.set_option("scratch_poison", poison)
            s = jr.miniapps.random_fields3d(ni, seed=3, iterMax=45, nout=15, bcs=bcs, dt=dt)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            st, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
            r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs, handle=h)
            outs.append((r, download_stokes(st)))
        finally:
            h.close()
    (ra, a), (rb, b) = outs
    assert ra.iter == rb.iter and list(ra.err_evo1) == list(rb.err_evo1), (list(ra.err_evo1), list(rb.err_evo1))
    for k in a:
        m = checks.interior_mask3d(k, a[k].shape)
        assert np.array_equal(a[k][m], b[k][m], equal_nan=True), k
