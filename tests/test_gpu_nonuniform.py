"""GPU parity on non-uniform grids (Geometry(xvi...), src/grid/Cartesian.jl:77-100) of the 2D drivers: visco-elastic solve!, multiphase visco-elasto-plastic
solve! (the path of miniapps/benchmarks/stokes2D/shear_band/ShearBand2D_refined.jl) and the single-phase non-linear driver, against the oracle's
spacing-array form; uniform spacing arrays reproduce the scalar path bit for bit; what is not built refuses."""
import numpy as np
import pytest

from test_oracle_nonuniform import _cp, _uniform_spacing, stretched

pytestmark = pytest.mark.gpu


def _grid(jr, ni, kx=2.0, ky=1.2):
    return jr.Geometry.from_vertices((stretched(ni[0], 0.0, 1.0, kx), stretched(ni[1], 0.0, 1.0, ky)))


@pytest.mark.parametrize("ni", [(33, 17), (130, 70)])
def test_visco_elastic_solve_on_a_stretched_grid(jr, oracle, ni):
    from justrelax_jl_amd import checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    s = jr.miniapps.random_fields2d(ni, iterMax=20, nout=5)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    g = _grid(jr, ni)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_solve(ref, oracle.set_spacing2d(checks.oracle_params2d(oracle, s), g.inv_spacing2d_host()))
    uni = _cp(s.arrays)
    oracle.stokes2d_solve(uni, checks.oracle_params2d(oracle, s))
    assert checks.max_rel_diff(ref["txx"], uni["txx"]) > 1e-2            # the stretched grid is a different problem
    stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    r = jr.solve_(stokes, s.pt, g, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 21
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-10)
    d = checks.compare_stokes(download_stokes(stokes), ref)
    assert max(d.values()) <= 1e-9, d


def test_uniform_spacing_arrays_equal_the_scalar_path_on_the_device(jr):
    from justrelax_jl_amd import checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    ni = (40, 24)
    s = jr.miniapps.random_fields2d(ni, iterMax=20, nout=5)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    outs = []
    for nonuni in (False, True):
        grid = s.grid
        if nonuni:      # a "non-uniform" Geometry whose vertices are equidistant
            grid = jr.Geometry.from_vertices(tuple(np.asarray(x) for x in s.grid.xvi))
            d = s.grid._di["center"]
            grid._di = dict(vertex=(np.full(ni[0], d[0]), np.full(ni[1], d[1])), center=(np.full(ni[0] - 1, d[0]), np.full(ni[1] - 1, d[1])),
                            velocity=((np.full(ni[0], d[0]), np.full(ni[1] + 1, d[1])), (np.full(ni[0] + 1, d[0]), np.full(ni[1], d[1]))))
        stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
        import ctypes as C
        from justrelax_jl_amd import _lib
        _lib.default_handle(0).call("jrx_tuning_set", C.c_char_p(b"fused2d"), C.c_int64(0))       # the same kernel form on both sides
        try:
            jr.solve_(stokes, s.pt, grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=s.kwargs)
        finally:
            _lib.default_handle(0).call("jrx_tuning_set", C.c_char_p(b"fused2d"), C.c_int64(1))
        outs.append(download_stokes(stokes))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k


def test_vep_solve_on_a_refined_grid(jr, oracle):
    """the shear band of test/test_shearband2D.jl on a grid refined towards the inclusion in x (ShearBand2D_refined.jl:205-210)"""
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _upload, _vep_params
    s = jr.miniapps.shearband2d(24, iterMax=59, nout=20)
    s.kwargs.update(iterMin=10)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    g = jr.Geometry.from_vertices((stretched(24, 0.0, 1.0, 1.8), np.linspace(0.0, 1.0, 25)))
    rng = np.random.default_rng(8)                                      # a pre-stress near the yield surface, so that cells and vertices yield at once
    for c in ("xx", "yy", "xy", "xy_c"):
        s.arrays["to" + c][...] = rng.uniform(-1.2, 1.2, size=s.arrays["to" + c].shape)
        s.arrays["t" + c][...] = s.arrays["to" + c]
    ref = _cp(s.arrays)
    rh = oracle.rheology_struct(s.extra["phases"])
    r_ref = oracle.stokes2d_vep_solve(ref, rh, oracle.set_spacing2d(_vep_params(oracle, s, iterMin=10), g.inv_spacing2d_host()))
    st, pr, ρg = _upload(jr, s)
    r = jr.solve_(st, s.pt, g, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"]
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-8)
    out = _download(jr, st)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-8, k
    assert (ref["eplxx"] != 0).any()                                    # plastic cells are present
    # strain_increment is not built on such a grid
    s.kwargs.update(strain_increment=True)
    with pytest.raises(RuntimeError, match="strain_increment on a non-uniform grid"):
        jr.solve_(st, s.pt, g, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)


def test_single_phase_driver_on_a_stretched_grid(jr, oracle):
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import VEP_MAP, _get
    from test_gpu_vep_extras import _nl_params
    import torch
    s = jr.miniapps.thermal_convection2d(32, ar=1, iterMax=99, nout=50)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    lx, ly = s.extra["li"]
    g = jr.Geometry.from_vertices((stretched(32, 0.0, lx, 1.5), stretched(32, -ly, 0.0, 1.0)))
    ph = dict(s.extra["rheology"])
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_nonlinear_solve(ref, oracle.rheology_struct([ph]), oracle.set_spacing2d(_nl_params(oracle, s), g.inv_spacing2d_host()))
    dev = torch.device("cuda", torch.cuda.current_device())
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP_MAP.items():
        _get(st, path).copy_(from_numpy(s.arrays[k], dev))
    ρg = (from_numpy(s.arrays["fx"], dev), from_numpy(s.arrays["fy"], dev))
    r = jr.solve_(st, s.pt, g, s.flow_bcs, ρg, ph, dict(T=from_numpy(s.arrays["T"], dev), P=st.P), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 100
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-8)
    for k, path in VEP_MAP.items():
        assert max_rel_diff(jr.to_numpy(_get(st, path)), ref[k]) <= 1e-8, k


def test_what_is_not_built_refuses(jr):
    s3 = jr.miniapps.random_fields3d((16, 9, 11), iterMax=2, nout=1)
    from justrelax_jl_amd.miniapps.common import upload_stokes
    stokes, ρg, K, G = upload_stokes(s3, jr.AMDGPUBackend)
    g3 = jr.Geometry.from_vertices(tuple(stretched(n) for n in s3.ni))
    with pytest.raises(NotImplementedError, match="2D drivers only"):
        jr.solve_(stokes, s3.pt, g3, s3.flow_bcs, ρg, K, G, s3.dt, None, kwargs=s3.kwargs)
    # the C ABI wants all six arrays or none
    import ctypes as C
    from justrelax_jl_amd import _lib, stokes as S
    s = jr.miniapps.random_fields2d((33, 17), iterMax=2, nout=1)
    st2, ρg2, K2, G2 = upload_stokes(s, jr.AMDGPUBackend)
    p = S.params2d(st2, s.pt, s.grid, s.flow_bcs, s.dt, iterMax=2, nout=1)
    p.inv_spacing[0] = st2.P.data_ptr()
    f = S.fields2d(st2, ρg2, K2, G2)
    res = _lib.SolveResult()
    with pytest.raises(_lib.JrxError, match="all six inverse-spacing arrays"):
        _lib.default_handle(0).call("jrx_stokes2d_solve", C.byref(f), C.byref(p), C.byref(res))


@pytest.mark.parametrize("form", ["rheology", "phases"])
def test_heat_diffusion_on_a_stretched_grid(jr, oracle, form):
    """heatdiffusion_PT! (rheology form, and the phase-ratio form GlobalConvection2D_DYREL_refined.jl calls on its refined grid) with the spacing vectors vs the oracle"""
    from justrelax_jl_amd.checks import max_rel_diff
    if form == "rheology":
        from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
        from test_gpu_stokes2d_thermal import _thermal_setup
        import justrelax_jl_amd.thermal as th
        s = jr.miniapps.diffusion2d(24, iterMax=300, nout=100)
        b = s.flow_bcs
        nx, ny = s.ni
        g = jr.Geometry.from_vertices((stretched(nx, 0.0, 100e3, 1.2), stretched(ny, -100e3, 0.0, 1.6)))
        p = oracle.thermal_params2d(s.ni, s.grid._di["center"], s.dt, 1e-30, iterMax=300, nout=100, no_flux=b.no_flux, constant_value=b.constant_value,
                                    constant_flux=b.constant_flux, periodic=b.periodic, rheology=s.extra["rheology"])
        oracle.thermal_bcs2d(s.arrays["T"], p)
        add_perturbation(s.arrays["T"], s.grid, **s.extra["perturbation"])
        thermal, pt, K, ρCp = _thermal_setup(jr, th, s)
        pt.ϵ = 1e-30
        ref = _cp(s.arrays)
        r_ref = oracle.heatdiffusion_PT2d(ref, oracle.set_spacing_thermal2d(p, g._di))
        uni = _cp(s.arrays)
        oracle.heatdiffusion_PT2d(uni, oracle.thermal_params2d(s.ni, s.grid._di["center"], s.dt, 1e-30, iterMax=300, nout=100, no_flux=b.no_flux,
                                                                constant_value=b.constant_value, constant_flux=b.constant_flux, periodic=b.periodic,
                                                                rheology=s.extra["rheology"]))
        assert max_rel_diff(ref["T"], uni["T"]) > 1e-4                      # a different problem than the uniform grid
        r = jr.heatdiffusion_PT_(thermal, pt, b, s.extra["rheology"], None, s.dt, g, kwargs=dict(iterMax=300, nout=100, verbose=False))
        assert list(r.iter_count) == list(r_ref["iter_count"]) == [100, 200, 300]
        assert np.allclose(r.norm_ResT, r_ref["norm_ResT"], rtol=1e-9)
        for name, t in (("T", thermal.T), ("qTx", thermal.qTx), ("qTy2", thermal.qTy2), ("ResT", thermal.ResT)):
            assert max_rel_diff(jr.to_numpy(t), ref[name]) <= 1e-9, name
        # the array form is refused on such a grid
        with pytest.raises(RuntimeError, match="array form"):
            jr.heatdiffusion_PT_(thermal, pt, b, K, ρCp, s.dt, g, kwargs=dict(iterMax=3, nout=1, verbose=False))
        return
    from test_gpu_thermal_multiphase import _device_setup, _oracle_inputs, _randomise
    s = jr.miniapps.diffusion2d_multiphase((37, 21), iterMax=60, nout=20)
    _randomise(s, 5)
    nx, ny = s.ni
    (x0, x1), (y0, y1) = [(float(v[0]), float(v[-1])) for v in s.grid.xvi]
    g = jr.Geometry.from_vertices((stretched(nx, x0, x1, 1.4), stretched(ny, y0, y1, 1.1)))
    thermal, pt, pr, args = _device_setup(jr, s, eps=1e-30)
    ref = _cp(s.arrays)
    p, m, ph = _oracle_inputs(oracle, s, 1e-30, iterMax=60, nout=20)
    r_ref = oracle.heatdiffusion_PT_phases(ref, oracle.set_spacing_thermal2d(p, g._di), m, ph)
    r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], args, s.dt, g, kwargs=dict(phase=pr, iterMax=60, nout=20, verbose=False))
    assert list(r.iter_count) == list(r_ref["iter_count"])
    assert np.allclose(r.norm_ResT, r_ref["norm_ResT"], rtol=1e-8)
    for name, t in (("T", thermal.T), ("qTx", thermal.qTx), ("qTy", thermal.qTy), ("ResT", thermal.ResT)):
        assert max_rel_diff(jr.to_numpy(t), ref[name]) <= 1e-9, name


def test_refined_shear_band_first_step(jr):
    """ShearBand2D_refined.jl through solve!: one time step on vertices refined towards the inclusion.  The reference's own convergence assertion of the
    uniform case (test_shearband2D.jl:194: err < 1e-6) holds, and the stress level is the visco-elastic build-up
    2 ε η (1 - exp(-G t / η)) of the matrix (Elastic_BuildUp.jl:4) whatever the spacing"""
    import math
    from test_gpu_vep2d import _upload
    n = 32
    xv = stretched(n, 0.0, 1.0, 1.8)
    s = jr.miniapps.shearband2d(n, iterMax=50_000, nout=500, xvi=(xv, np.linspace(0.0, 1.0, n + 1)))
    assert s.grid.nonuniform and np.ptp(np.diff(xv)) > 0.5 * np.diff(xv).min()
    st, pr, ρg = _upload(jr, s)
    r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    assert r.err_evo1[-1] < 1.0e-6 and r.iter < 50_000
    jr.tensor_invariant_(st.τ)
    τII = jr.to_numpy(st.τ.II)
    want = 2.0 * 1.0 * 1.0 * (1.0 - math.exp(-1.0 * s.dt / 1.0))
    # (the weak inclusion unloads the corners by a few per cent; the largest stress in the box stays within 6 % of the build-up value at this resolution (0.4 % at n = 64) -- the uniform-grid
    # run of test/test_shearband2D_softening.jl:199-205 asserts the same 0.4423 at t = 0.25)
    assert τII.max() == pytest.approx(want, rel=6e-2) and τII.min() > 0.25 * want
    assert τII.max() < 1.6                                    # below yield after the first step
