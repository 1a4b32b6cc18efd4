"""GPU parity tests, 3D pseudo-transient heat diffusion (DiffusionPT_solver.jl with the 3D kernels) vs the CPU
oracle, and the reference's own 3D diffusion numbers (test/test_diffusion3D.jl:150-151) on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_ITERS = 1e-9


def _setup(jr, s):
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    for name in ("T", "Told", "H", "qTx", "qTy", "qTz", "qTx2", "qTy2", "qTz2", "shear_heating"):
        getattr(thermal, name).copy_(from_numpy(s.arrays[name], dev))
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=s.pt["eps"])
    return thermal, pt, K, ρCp


def _params(orc, s, eps, **kw):
    b = s.flow_bcs
    return orc.thermal_params3d(s.ni, s.grid._di["center"], s.dt, eps, no_flux=b.no_flux, constant_value=b.constant_value,
                                constant_flux=b.constant_flux, periodic=b.periodic, **kw)


def _face_mask(shape):
    """ghost edges/corners of T are written by several statements of the BC kernels in the reference (later wins) and never read"""
    g = np.zeros(shape, dtype=int)
    for d in range(3):
        idx = [slice(None)] * 3
        for e in (0, shape[d] - 1):
            idx[d] = e
            g[tuple(idx)] += 1
    return g < 2


def test_thermal_bcs_3d_all_kinds(jr, oracle):
    import torch
    from justrelax_jl_amd import thermal as th
    from justrelax_jl_amd.arrays import from_numpy
    rng = np.random.default_rng(1)
    T0 = np.asfortranarray(rng.standard_normal((11, 8, 9)))
    dev = torch.device("cuda", torch.cuda.current_device())
    cases = [jr.TemperatureBoundaryConditions(no_flux=dict(left=True, right=True, front=True, back=True, top=False, bot=False),
                                              constant_value=dict(left=True, right=True, front=True, back=True, top=300.0, bot=3500.0)),
             jr.TemperatureBoundaryConditions(no_flux=dict(left=False, right=False, front=False, back=False, top=True, bot=False),
                                              constant_value=dict(left=5.0, right=False, front=False, back=-2.0, top=False, bot=1.5),
                                              periodic=dict(left=False, right=False, front=False, back=False, top=False, bot=False)),
             jr.TemperatureBoundaryConditions(no_flux=dict(left=False, right=False, front=False, back=False, top=False, bot=False),
                                              periodic=dict(left=True, right=True, front=True, back=True, top=True, bot=True))]
    for b in cases:
        p = oracle.thermal_params3d((9, 6, 7), (1.0, 1.0, 1.0), 1.0, 0.0, no_flux=b.no_flux, constant_value=b.constant_value,
                                    constant_flux=b.constant_flux, periodic=b.periodic)
        Tref = T0.copy(order="F")
        import ctypes as C
        oracle.lib().orc_thermal_bcs3d(Tref.ctypes.data_as(C.POINTER(C.c_double)), C.byref(p))
        Td = from_numpy(T0, dev)
        th.thermal_bcs_(Td, b)
        assert np.array_equal(jr.to_numpy(Td), Tref)            # same statement order -> ghost edges agree as well
        assert not np.array_equal(Tref, T0)


@pytest.mark.parametrize("form,ni", [("array", (20, 14, 12)), ("rheology", (20, 14, 12)), ("array", (70, 17, 20)), ("rheology", (130, 9, 35)),
                                     ("array", (256, 256, 256)), ("rheology", (192, 160, 128))])
def test_thermal3d_iterations_match_oracle(jr, oracle, form, ni):
    """jrx_heatdiffusion_PT3d (fused flux + update + BC kernel on the unobserved iterations, the two kernels in place on check / last
    iterations, ping-pong (T, qT) sets; the larger grids span several tiles and z chunks) against the oracle's loop"""
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.diffusion3d(ni, iterMax=300, nout=100)
    rheo = s.extra["rheology"] if form == "rheology" else None
    p = _params(oracle, s, 1e-30, iterMax=300, nout=100, rheology=rheo)
    thermal, pt, K, ρCp = _setup(jr, s)
    pt.ϵ = 1e-30
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r_ref = oracle.heatdiffusion_PT3d(ref, p)
    A, B = (rheo, None) if form == "rheology" else (K, ρCp)
    r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, A, B, s.dt, s.grid, kwargs=dict(iterMax=300, nout=100, verbose=False))
    assert list(r.iter_count) == list(r_ref["iter_count"]) == [100, 200, 300]
    assert np.allclose(r.norm_ResT, r_ref["norm_ResT"], rtol=1e-9)
    for name, t in (("T", thermal.T), ("Told", thermal.Told), ("dT", thermal.ΔT)):
        got = jr.to_numpy(t)          # whole arrays: the in-kernel BC replay must reproduce the ghost edges and corners as well
        assert np.abs(got - ref[name]).max() <= TOL_ITERS * np.abs(ref[name]).max(), name
    for name, t in (("qTx", thermal.qTx), ("qTy", thermal.qTy), ("qTz", thermal.qTz), ("qTz2", thermal.qTz2)):
        assert max_rel_diff(jr.to_numpy(t), ref[name]) <= TOL_ITERS, name
    # the residual of a converged run is rounding noise of its terms (H = 1e-6 among them): absolute comparison on that scale
    assert np.abs(jr.to_numpy(thermal.ResT) - ref["ResT"]).max() <= TOL_ITERS * max(np.abs(ref["ResT"]).max(), 1e-6)


def test_diffusion3d_reference_numbers_on_the_gpu(jr, oracle):
    """test/test_diffusion3D.jl:143-151 on the device: 32^3, 10 steps of 50 kyr; T[16,16,16] ≈ 1813.2470160788096,
    T[17,17,17] ≈ 1831.2568044653274 (reference tolerance rtol 1e-3; observed agreement ~1e-15)"""
    s = jr.miniapps.diffusion3d(32)
    thermal, pt, K, ρCp = _setup(jr, s)
    for _ in range(s.extra["nt"]):
        r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], None, s.dt, s.grid, kwargs=dict(verbose=False))
        assert r.norm_ResT[-1] <= 1e-8
    T = jr.to_numpy(thermal.T)
    assert T[15, 15, 15] == pytest.approx(1813.2470160788096, rel=1.0e-12)
    assert T[16, 16, 16] == pytest.approx(1831.2568044653274, rel=1.0e-12)


@pytest.mark.parametrize("ni", [(256, 256, 256), (200, 96, 70)])
def test_fused_and_two_kernel_iterations_agree_at_full_size(jr, ni):
    """BASELINE-size property check (no oracle at this size): 60 iterations of jrx_heatdiffusion_PT3d with the fused one-launch
    iteration (ping-pong (T, qT) sets, BC ghost replay in-kernel) and with compute_flux! / update_T! as two launches give the same
    T, qT and residual history bit for bit -- same arithmetic in the same order."""
    import ctypes as C
    from justrelax_jl_amd import _lib
    s = jr.miniapps.diffusion3d(ni, iterMax=60, nout=20)
    h = _lib.default_handle()
    outs = []
    try:
        for fused in (1, 0):
            h.call("jrx_set_option", C.c_char_p(b"thermal_fused"), C.c_int64(fused))
            thermal, pt, K, ρCp = _setup(jr, s)
            pt.ϵ = 1e-30
            r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=60, nout=20, verbose=False))
            outs.append((list(r.iter_count), list(r.norm_ResT), {k: jr.to_numpy(getattr(thermal, k)) for k in ("T", "qTx", "qTy", "qTz", "ResT")}))
            del thermal, pt, K, ρCp
    finally:
        h.call("jrx_set_option", C.c_char_p(b"thermal_fused"), C.c_int64(1))
    assert outs[0][0] == outs[1][0] == [20, 40, 60]
    assert outs[0][1] == outs[1][1]
    for k in outs[0][2]:
        assert np.array_equal(outs[0][2][k], outs[1][2][k]), k
    T = outs[0][2]["T"]
    assert np.isfinite(T).all() and T.max() > 1500.0


@pytest.mark.parametrize("tile", [4, 8])
@pytest.mark.parametrize("form,ni", [("array", (70, 17, 20)), ("rheology", (130, 9, 35)), ("array", (200, 96, 70)), ("rheology", (64, 21, 13))])
def test_tiled_fused_iteration_keeps_the_bits(jr, form, ni, tile):
    """round 6, tuning switch "thermal_tile": the one-launch iteration on 64 x TY tiles whose rows exchange (T, K, θ) and the new y flux through LDS (k_thermal3d_fused_t) gives
    the bits of the row-segment form and of the two-kernel iteration -- grids whose ny fills whole tiles, leaves a partial last tile (17, 21, 9 rows) and whose nx is not a
    multiple of 64"""
    from justrelax_jl_amd import _lib
    s = jr.miniapps.diffusion3d(ni, iterMax=60, nout=20)
    rheo = s.extra["rheology"] if form == "rheology" else None
    h = _lib.default_handle()
    outs = []
    try:
        for fused, tl in ((1, tile), (1, 0), (0, 0)):
            h.set_option("thermal_fused", fused)
            h.set_option("thermal_tile", tl)
            thermal, pt, K, ρCp = _setup(jr, s)
            pt.ϵ = 1e-30
            A, B = (rheo, None) if form == "rheology" else (K, ρCp)
            n0 = h.get_option("stat_thermal_fused")
            r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, A, B, s.dt, s.grid, kwargs=dict(iterMax=60, nout=20, verbose=False))
            assert (h.get_option("stat_thermal_fused") > n0) == bool(fused)
            outs.append((list(r.iter_count), list(r.norm_ResT), {k: jr.to_numpy(getattr(thermal, k)) for k in ("T", "qTx", "qTy", "qTz", "ResT")}))
            del thermal, pt, K, ρCp
    finally:
        h.set_option("thermal_fused", 1)
        h.set_option("thermal_tile", 0)
    for v in (1, 2):
        assert outs[0][0] == outs[v][0] == [20, 40, 60] and outs[0][1] == outs[v][1]
        for k in outs[0][2]:
            assert np.array_equal(outs[0][2][k], outs[v][2][k]), (v, k)
