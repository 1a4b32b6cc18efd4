"""Every BASELINE.json config at its stated size, HIP path (C ABI) against the CPU oracle on the same inputs.

The oracle is OpenMP C: on the GPU box's host cores a bounded number of PT iterations at 256^3 / 512^2 / 1024^2 takes seconds, so
these are direct parity tests at the sizes the bench quotes (the kernel choice is size-dependent: the auto rule picks k_fused3d at
256^3, the two-kernel 2D loop above 200^2 nodes), not only size-independent properties:

  configs[2]  SolVi3D 256^3 (miniapps/benchmarks/stokes3D/solvi/SolVi3D.jl:45-129)  -- jrx_stokes3d_solve, 25 iterations, 2 checks
  (8d)        random fields, finite dt, G, K at 256^3 (SURVEY 8d kernel-parity inputs)      -- jrx_stokes3d_solve, 17 iterations, general k_fused3d
  configs[3]  SolVi3D 512^3 per GPU (the headline's block)                          -- jrx_stokes3d_solve, 9 iterations, 2 checks (one 48 GB host copy)
  configs[1]  SolCx 512^2 (miniapps/benchmarks/stokes2D/solcx/SolCx.jl:54-116)      -- jrx_stokes2d_solve, 200 iterations vs oracle,
              then the reference's convergence assertion err_evo1[end] < 1e-8 (test/test_stokes_solcx.jl:26-37) at full size
  configs[4]  shear band 1024^2 (test/test_shearband2D.jl:61-145)                   -- jrx_stokes2d_vep_solve, 61 iterations
  configs[0]  thermal diffusion 256^2 (test/test_diffusion2D.jl:46-96)              -- jrx_heatdiffusion_PT2d, one time step

fp64 tolerance: 1e-9 relative after tens of iterations (observed: bit-identical or ~1e-13)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cp(arrs):
    return {k: v.copy(order="F") for k, v in arrs.items()}


def _stat(h, key):
    v = C.c_int64(0)
    h.call("jrx_get_option", C.c_char_p(key.encode()), C.byref(v))
    return v.value


def test_solvi3d_256_matches_oracle(jr, oracle):
    import torch
    from justrelax_jl_amd import _lib, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    n = 256
    s = jr.miniapps.solvi3d(n, iterMax=24, nout=10)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    ref = _cp(s.arrays)
    r_ref = oracle.stokes3d_solve(ref, checks.oracle_params3d(oracle, s))
    stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    h = _lib.default_handle()
    before = _stat(h, "stat_fused3d")
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
    assert _stat(h, "stat_fused3d") > before, "the auto rule did not select k_fused3d at 256^3"
    assert r.iter == r_ref["iter"] == 25
    for k in ("norm_Rx", "norm_Ry", "norm_Rz", "norm_divV", "err_evo1"):
        assert np.allclose(getattr(r, k), r_ref[k], rtol=1e-10, atol=0), k
    dev = download_stokes(stokes)
    names = ["P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "toxx", "toxy", "Rx", "Ry", "Rz", "RP", "divV", "exx", "eyz"]
    d = checks.compare_stokes(dev, ref, names)
    assert max(d.values()) <= 1e-9, d
    del stokes
    torch.cuda.empty_cache()


def test_random_fields_256_finite_dt_matches_oracle_general_kernel(jr, oracle):
    """SURVEY 8(d)'s kernel-parity inputs at a BASELINE size: every field ~U(-1,1) (default_rng(20260821)), eta = 10^U(-3,0), G ~ 1, K ~ 2, dt = 0.25, Q ~ U(-0.1,0.1)
    at 256^3 -- SolVi3D zeroes every elastic / compressible term (dt = Inf, F7), this run does not: the elastic increment of compute_tau! (StressKernels.jl:2-5,149-230:
    (tau - tau_o) eta / (G dt), the clamped 4-cell means of eta and G at the shear nodes) and the compressible pressure update (PressureKernels.jl:186-195: P0 / (K dt),
    Q / dt, psi / (K dt)) are compared with the oracle through jrx_stokes3d_solve, and the launch counters prove that the GENERAL form of k_fused3d ran (none of its
    launches in the viscous-limit form)."""
    import torch
    from justrelax_jl_amd import _lib, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    n = 256
    s = jr.miniapps.random_fields3d((n, n, n), seed=20260821, dt=0.25, G=1.0, K=2.0, iterMax=16, nout=8)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    ref = _cp(s.arrays)
    r_ref = oracle.stokes3d_solve(ref, checks.oracle_params3d(oracle, s))
    stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    h = _lib.default_handle()
    f0, v0 = _stat(h, "stat_fused3d"), _stat(h, "stat_fused3d_visc")
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
    assert _stat(h, "stat_fused3d") - f0 >= 8, "the auto rule did not select k_fused3d at 256^3"
    assert _stat(h, "stat_fused3d_visc") == v0, "a finite-dt run launched the viscous-limit form"
    assert r.iter == r_ref["iter"] == 17
    for k in ("norm_Rx", "norm_Ry", "norm_Rz", "norm_divV", "err_evo1"):
        assert np.allclose(getattr(r, k), r_ref[k], rtol=1e-10, atol=0), k
    dev = download_stokes(stokes)
    names = ["P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "toxx", "toyy", "tozz", "toyz", "toxz", "toxy", "Rx", "Ry", "Rz", "RP", "divV", "exx", "eyy", "ezz", "eyz", "exz", "exy"]
    d = checks.compare_stokes(dev, ref, names)
    assert max(d.values()) <= 1e-9, d
    del stokes, dev
    torch.cuda.empty_cache()


def test_solvi3d_512_matches_oracle(jr, oracle):
    """configs[3]'s block size (the headline's): 9 iterations of jrx_stokes3d_solve at 512^3 against the oracle.  Memory: one host copy of the 45 fields
    (48 GB) -- it is uploaded first, then the oracle advances it in place, and the device result is brought back one field at a time."""
    import torch
    from justrelax_jl_amd import _lib, checks
    from justrelax_jl_amd.arrays import to_numpy
    from justrelax_jl_amd.miniapps.common import upload_stokes, _get, stokes_field_names
    n = 512
    s = jr.miniapps.solvi3d(n, iterMax=8, nout=4)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    h = _lib.default_handle()
    before = _stat(h, "stat_fused3d")
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
    assert _stat(h, "stat_fused3d") > before
    r_ref = oracle.stokes3d_solve(s.arrays, checks.oracle_params3d(oracle, s))        # in place on the host copy
    assert r.iter == r_ref["iter"] == 9
    for k in ("norm_Rx", "norm_Ry", "norm_Rz", "norm_divV", "err_evo1"):
        assert np.allclose(getattr(r, k), r_ref[k], rtol=1e-10, atol=0), k
    paths = stokes_field_names(3)
    for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "Rx", "Rz", "RP"):
        got = to_numpy(_get(stokes, paths[k]))
        d = checks.compare_stokes({k: got}, {k: s.arrays[k]}, [k])
        assert d[k] <= 1e-9, (k, d[k])
        del got
    del stokes, ρg, K, G
    torch.cuda.empty_cache()


def test_solcx_512_matches_oracle_and_converges(jr, oracle):
    from justrelax_jl_amd import _lib, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    n = 512
    s = jr.miniapps.solcx2d(n, iterMax=199, nout=50)
    eps = (s.pt.ϵ_rel, s.pt.ϵ_abs)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_solve(ref, checks.oracle_params2d(oracle, s))
    stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    h = _lib.default_handle()
    before = _stat(h, "stat_fused2d")
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
    assert _stat(h, "stat_fused2d") > before           # 512^2 runs the one-launch iteration (the form with batched loads; up to 1.2 M nodes)
    assert r.iter == r_ref["iter"] == 200
    for k in ("norm_Rx", "norm_Ry", "norm_divV", "err_evo1"):
        assert np.allclose(getattr(r, k), r_ref[k], rtol=1e-10, atol=0), k
    d = checks.compare_stokes(download_stokes(stokes), ref)
    assert max(d.values()) <= 1e-9, d
    # the two-kernel loop, forced at this size, leaves the same bits
    try:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(2))
        before = _stat(h, "stat_fused2d")
        st3, ρg3, K3, G3 = upload_stokes(s, jr.AMDGPUBackend)
        r3 = jr.solve_(st3, s.pt, s.grid, s.flow_bcs, ρg3, G3, K3, s.dt, None, kwargs=s.kwargs)
        assert _stat(h, "stat_fused2d") == before
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
    a, b = download_stokes(stokes), download_stokes(st3)
    for k in ("P", "txx", "tyy", "txy", "Rx", "Ry"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("Vx", "Vy"):
        assert np.array_equal(a[k][1:-1, 1:-1], b[k][1:-1, 1:-1]), k
    assert r3.iter == r.iter and np.array_equal(r3.err_evo1, r.err_evo1)
    # the reference's assertion at the BASELINE size: err_evo1[end] < 1e-8 with the script's own tolerances and iterMax
    s.pt.ϵ_rel, s.pt.ϵ_abs = eps
    stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=dict(iterMax=500_000, nout=5000, verbose=False))
    assert r.err_evo1[-1] < 1.0e-8, (r.iter, r.err_evo1[-1])
    assert r.iter < 500_000


def test_shearband_1024_matches_oracle(jr, oracle):
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _upload, _vep_params
    s = jr.miniapps.shearband2d(1024, iterMax=60, nout=20)
    s.kwargs.update(iterMin=10)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]), _vep_params(oracle, s, iterMin=10))
    stokes, pr, ρg = _upload(jr, s)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 61
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-9, k


def test_thermal2d_256_matches_oracle(jr, oracle):
    """one 50-kyr step of test_diffusion2D.jl's set-up at 256^2 (array-coefficient form, fused one-launch iterations)"""
    from justrelax_jl_amd import _lib, thermal as th
    from justrelax_jl_amd.checks import max_rel_diff
    from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
    from test_gpu_stokes2d_thermal import _thermal_setup
    n, iters = 256, 3000
    s = jr.miniapps.diffusion2d(n, iterMax=iters, nout=1000)
    b = s.flow_bcs
    p = oracle.thermal_params2d(s.ni, s.grid._di["center"], s.dt, 1e-300, iterMax=iters, nout=1000, no_flux=b.no_flux,
                                constant_value=b.constant_value, constant_flux=b.constant_flux, periodic=b.periodic)
    oracle.thermal_bcs2d(s.arrays["T"], p)
    add_perturbation(s.arrays["T"], s.grid, **s.extra["perturbation"])
    thermal, pt, K, ρCp = _thermal_setup(jr, th, s)
    pt.ϵ = 1e-300
    ref = _cp(s.arrays)
    r_ref = oracle.heatdiffusion_PT2d(ref, p)
    h = _lib.default_handle()
    before = _stat(h, "stat_thermal_fused")
    r = jr.heatdiffusion_PT_(thermal, pt, b, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=iters, nout=1000, verbose=False))
    assert _stat(h, "stat_thermal_fused") > before
    assert list(r.iter_count) == list(r_ref["iter_count"]) == [1000, 2000, 3000]
    assert np.allclose(r.norm_ResT, r_ref["norm_ResT"], rtol=1e-9)
    for name, t in (("T", thermal.T), ("Told", thermal.Told), ("dT", thermal.ΔT), ("qTx", thermal.qTx), ("qTy2", thermal.qTy2), ("ResT", thermal.ResT)):
        assert max_rel_diff(jr.to_numpy(t), ref[name]) <= 1e-9, name
