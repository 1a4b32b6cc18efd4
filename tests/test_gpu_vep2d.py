"""GPU parity tests, 2D multiphase visco-elasto-plastic Stokes (config 5, Stokes2D.jl:577-866) vs the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VEP_MAP = dict(P="P", P0="P0", divV="divV", Q="Q", Vx="V.Vx", Vy="V.Vy", Ux="U.Ux", Uy="U.Uy", exx="ε.xx", eyy="ε.yy", exy="ε.xy", exy_c="ε.xy_c",
               eplxx="ε_pl.xx", eplyy="ε_pl.yy", eplxy="ε_pl.xy", eplxy_c="ε_pl.xy_c", txx="τ.xx", tyy="τ.yy", txy="τ.xy", txy_c="τ.xy_c",
               tII="τ.II", toxx="τ_o.xx", toyy="τ_o.yy", toxy="τ_o.xy", toxy_c="τ_o.xy_c", eta="viscosity.η", eta_v="viscosity.ηv",
               eta_vep="viscosity.η_vep", EII_pl="EII_pl", evol_pl="ε_vol_pl", EVol_pl="EVol_pl", RP="R.RP", Rx="R.Rx", Ry="R.Ry", omega_xy="ω.xy")


def _get(o, path):
    for p in path.split("."):
        o = getattr(o, p)
    return o


def _upload(jr, s):
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP_MAP.items():
        _get(st, path).copy_(from_numpy(s.arrays[k], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    pr.center.copy_(from_numpy(s.arrays["phase_c"], dev))
    pr.vertex.copy_(from_numpy(s.arrays["phase_v"], dev))
    ρg = (from_numpy(s.arrays["fx"], dev), from_numpy(s.arrays["fy"], dev))
    return st, pr, ρg


def _download(jr, st):
    return {k: jr.to_numpy(_get(st, path)) for k, path in VEP_MAP.items()}


def _vep_params(orc, s, **over):
    pt, b = s.pt, s.flow_bcs
    kw = dict(iterMax=s.kwargs["iterMax"], nout=s.kwargs["nout"], stag_mode=1)
    kw.update(over)
    return orc.vep_params2d(s.ni, s.grid._di["center"], s.dt, dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                            free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, **kw)


def _randomize(s, seed=4):
    """make every input of the stress kernel non-trivial: yielding and non-yielding nodes, mixed phase ratios"""
    rng = np.random.default_rng(seed)
    a = s.arrays
    for k in ("P", "exx", "eyy", "exy", "txx", "tyy", "txy", "txy_c", "toxx", "toyy", "toxy", "toxy_c"):
        a[k][...] = rng.uniform(-2.0, 2.0, size=a[k].shape)
    a["eta"][...] = 10.0 ** rng.uniform(-1.0, 0.5, size=a["eta"].shape)
    for k in ("phase_c", "phase_v"):
        r = rng.uniform(0.0, 1.0, size=a[k].shape[1:])
        r[rng.uniform(size=r.shape) < 0.3] = 0.0
        r[rng.uniform(size=r.shape) < 0.3] = 1.0
        a[k][0], a[k][1] = r, 1.0 - r


def test_update_stresses_matches_oracle(jr, oracle):
    """update_stresses_center_vertex_ps! (StressKernels.jl:992-1144) on random states: τ, λ, ε_pl, τII, η_vep, Pr_c"""
    from justrelax_jl_amd import _lib, stokes as st_mod
    from justrelax_jl_amd.arrays import from_numpy
    import torch
    s = jr.miniapps.shearband2d(24)
    _randomize(s)
    rh = oracle.rheology_struct(s.extra["phases"])
    p = _vep_params(oracle, s)
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    rng = np.random.default_rng(9)
    theta = np.asfortranarray(rng.uniform(-1, 1, size=s.ni))
    lam = np.asfortranarray(rng.uniform(0, 0.1, size=s.ni))
    lamv = np.asfortranarray(rng.uniform(0, 0.1, size=(s.ni[0] + 1, s.ni[1] + 1)))
    lam_r, lamv_r = lam.copy(order="F"), lamv.copy(order="F")
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    f = oracle.vep2d(ref)
    oracle.lib().orc_vep2d_stress(C.byref(f), dp(theta), dp(lam_r), dp(lamv_r), C.byref(rh), C.byref(p))
    stokes, pr, ρg = _upload(jr, s)
    dev = stokes.P.device
    th_d, lam_d, lamv_d = from_numpy(theta, dev), from_numpy(lam, dev), from_numpy(lamv, dev)
    h = _lib.default_handle()
    fd = st_mod.vep_fields2d(stokes, ρg, pr)
    pd = st_mod.vep_params2d(stokes, s.pt, s.grid, s.flow_bcs, s.dt)
    h.call("jrx_vep2d_update_stresses", C.byref(fd), C.c_void_p(th_d.data_ptr()), C.c_void_p(lam_d.data_ptr()), C.c_void_p(lamv_d.data_ptr()),
           C.byref(st_mod.rheology_table(s.extra["phases"])), C.byref(pd))
    dev_out = _download(jr, stokes)
    from justrelax_jl_amd.checks import max_rel_diff
    assert (lam_r != lam).any() and (ref["eplxy"] != 0).any() and (ref["eplxy"] == 0).any()      # both branches exercised
    for k in ("txx", "tyy", "txy", "txy_c", "tII", "eta_vep", "P", "eplxx", "eplyy", "eplxy", "evol_pl"):
        assert max_rel_diff(dev_out[k], ref[k]) <= 1e-12, k
    assert max_rel_diff(jr.to_numpy(lam_d), lam_r) <= 1e-12 and max_rel_diff(jr.to_numpy(lamv_d), lamv_r) <= 1e-12


def test_vep_solve_matches_oracle_over_iterations(jr, oracle):
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.shearband2d(24, iterMax=60, nout=20)
    s.kwargs.update(iterMin=10)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]), _vep_params(oracle, s, iterMin=10))
    stokes, pr, ρg = _upload(jr, s)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 61
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-9, k


def test_shearband2d_reference_test(jr, oracle):
    """test/test_shearband2D.jl:194-202 on the GPU (and step by step against the oracle)."""
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.shearband2d(32)
    rh, p = oracle.rheology_struct(s.extra["phases"]), _vep_params(oracle, s)
    stokes, pr, ρg = _upload(jr, s)
    jr.compute_viscosity_(stokes, pr, None, s.extra["phases"], (-np.inf, np.inf))
    assert float(stokes.viscosity.η.min()) == float(stokes.viscosity.η.max()) == 1.0
    tII = []
    for it in range(10):
        iters = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
        r_ref = oracle.stokes2d_vep_solve(s.arrays, rh, p)
        assert iters.iter == r_ref["iter"], it
        jr.tensor_invariant_(stokes.ε)
        tII.append(float(stokes.τ.xx.max()))
        assert tII[-1] == pytest.approx(s.arrays["txx"].max(), rel=1e-7)
    assert iters.err_evo1[-1] < 1.0e-6
    jr.tensor_invariant_(stokes.τ)
    II = jr.to_numpy(stokes.τ.II)
    assert II.min() == pytest.approx(1.5128689768248313, abs=1.0e-3)
    assert II.max() == pytest.approx(1.6415759440014273, abs=1.0e-3)
    assert tII[-1] == pytest.approx(1.6376258215356436, abs=1.0e-4)
    assert max_rel_diff(jr.to_numpy(stokes.EII_pl), s.arrays["EII_pl"]) < 1e-6


@pytest.mark.parametrize("multiphase", [False, True])
def test_compute_tau_nonlinear_matches_oracle(jr, oracle, multiphase):
    """compute_τ_nonlinear! 2D (StressKernels.jl:266-351 + rheology/StressUpdate.jl:2-105) on random states, both the
    single-phase and the phase-ratio form; yielding and non-yielding cells.  Tolerance 1e-12 (observed: bit-identical)."""
    from justrelax_jl_amd import stokes as st_mod
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.shearband2d(24)
    _randomize(s, seed=12)
    phases = [dict(ph) for ph in s.extra["phases"]]
    for ph in phases:
        ph["psi_deg"] = 7.0              # exercise the dilatant terms (volume, θ)
        ph["Kb"] = 3.0
    rh = oracle.rheology_struct(phases)
    p = _vep_params(oracle, s)
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    rng = np.random.default_rng(5)
    lam = np.asfortranarray(rng.uniform(0, 0.1, size=s.ni))
    lam_r = lam.copy(order="F")
    theta_r = np.zeros(s.ni, order="F")
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    f = oracle.vep2d(ref)
    oracle.lib().orc_compute_tau_nonlinear2d(C.byref(f), dp(theta_r), dp(lam_r), C.byref(rh), C.byref(p), C.c_int32(int(multiphase)))
    stokes, pr, ρg = _upload(jr, s)
    dev = stokes.P.device
    lam_d, th_d = from_numpy(lam, dev), jr.fzeros(s.ni, dev)
    st_mod.compute_τ_nonlinear_(stokes, th_d, lam_d, phases, s.dt, s.pt, phase_ratios=pr if multiphase else None)
    out = _download(jr, stokes)
    assert (lam_r != lam).any() and (lam_r == lam).any()                  # yielding and elastic cells
    for k in ("txx", "tyy", "txy_c", "tII", "eta_vep", "eplxx", "eplyy", "eplxy"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-12, k
    for k in ("txy", "P", "toxy", "exy"):                                  # untouched
        assert np.array_equal(out[k], s.arrays[k]), k
    assert max_rel_diff(jr.to_numpy(lam_d), lam_r) <= 1e-12 and max_rel_diff(jr.to_numpy(th_d), theta_r) <= 1e-12
    assert np.abs(theta_r - ref["P"]).max() > 0


@pytest.mark.parametrize("ni", [(24, 17), (3, 5), (130, 64)])
def test_center2vertex_matches_oracle(jr, oracle, ni):
    """center2vertex!(τ.xy, τ.xy_c) (Interpolations.jl:101-114), bit-exact incl. the edge copies"""
    import torch
    from justrelax_jl_amd import stokes as st_mod
    from justrelax_jl_amd.arrays import from_numpy
    rng = np.random.default_rng(2)
    c = np.asfortranarray(rng.standard_normal(ni))
    v = np.asfortranarray(rng.standard_normal((ni[0] + 1, ni[1] + 1)))
    v_ref = v.copy(order="F")
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    oracle.lib().orc_center2vertex2d(dp(v_ref), dp(c), C.c_int64(ni[0]), C.c_int64(ni[1]))
    dev = torch.device("cuda", torch.cuda.current_device())
    vd, cd = from_numpy(v, dev), from_numpy(c, dev)
    st_mod.center2vertex_(vd, cd)
    assert np.array_equal(jr.to_numpy(vd), v_ref)
    assert np.array_equal(v_ref[0, 1:-1], v_ref[1, 1:-1]) and np.array_equal(v_ref[:, 0], v_ref[:, 1])


def test_epilogue_operators_2d(jr, oracle):
    """shear2center!, accumulate_tensor!, compute_vorticity! 2D as stand-alone operators, vs numpy restatements (bit-exact)"""
    from justrelax_jl_amd import stokes as st_mod
    s = jr.miniapps.shearband2d(20)
    _randomize(s, seed=6)
    rng = np.random.default_rng(1)
    a = s.arrays
    for k in ("Vx", "Vy", "eplxx", "eplyy", "eplxy", "EII_pl"):
        a[k][...] = rng.uniform(-1, 1, size=a[k].shape)
    stokes, pr, ρg = _upload(jr, s)
    st_mod.shear2center_(stokes.ε)
    v = a["exy"]
    assert np.array_equal(jr.to_numpy(stokes.ε.xy_c), 0.25 * (v[:-1, :-1] + v[1:, :-1] + v[:-1, 1:] + v[1:, 1:]))
    st_mod.accumulate_tensor_(stokes.EII_pl, stokes.ε_pl, 0.25)
    II = oracle.tensor_invariant2d(a["eplxx"], a["eplyy"], a["eplxy"], mode=1)
    assert np.array_equal(jr.to_numpy(stokes.EII_pl), a["EII_pl"] + II * 0.25)
    st_mod.compute_vorticity_(stokes, s.grid)
    _dx, _dy = s.grid._di["center"]
    Vx, Vy = a["Vx"], a["Vy"]
    nx, ny = s.ni
    want = 0.5 * ((-Vy[:nx + 1, :ny + 1] + Vy[1:nx + 2, :ny + 1]) * _dx - (-Vx[:nx + 1, :ny + 1] + Vx[:nx + 1, 1:ny + 2]) * _dy)
    assert np.array_equal(jr.to_numpy(stokes.ω.xy), want)


@pytest.mark.parametrize("ni,iters,nout", [(32, 40, 10), (70, 37, 7), (63, 12, 5), (130, 45, 20), (256, 80, 40)])
def test_vep2d_batched_kernels_equal_the_control_flow_kernels(jr, ni, iters, nout):
    """option fused2d_batch: k_vep_pre_b and k_vep_visc_velocity_b (every operand requested up front; uniform grid, constant densities, viscosity laws without fields) against
    k_vep_pre and k_vep_visc_velocity, and option vep3_np_const (the stress kernel's instantiation with the phase count as a constant) against the run-time loops: every field
    of the solve bit for bit -- observed and unobserved iterations, graph replays at the small sizes, rows of one / several waves"""
    from justrelax_jl_amd import _lib
    h = _lib.default_handle()
    outs, res = [], []
    try:
        for batch, npc in ((0, 0), (1, 1), (1, 0), (0, 1)):
            h.set_option("fused2d_batch", batch)
            h.set_option("vep3_np_const", npc)
            s = jr.miniapps.shearband2d(ni, iterMax=iters - 1, nout=nout)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            stokes, pr, ρg = _upload(jr, s)
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
            outs.append(_download(jr, stokes))
            res.append(r)
    finally:
        h.set_option("fused2d_batch", 1)
        h.set_option("vep3_np_const", 1)
    assert res[0].iter == iters
    for v in (1, 2, 3):
        assert res[v].iter == res[0].iter and np.array_equal(np.asarray(res[v].err_evo1), np.asarray(res[0].err_evo1))
        for k in outs[0]:
            a, b = outs[0][k], outs[v][k]
            if k in ("Vx", "Vy", "Ux", "Uy"):      # the four ghost corners are not read by any stencil
                a, b = a.copy(), b.copy()
                for c in ((0, 0), (0, -1), (-1, 0), (-1, -1)):
                    a[c] = b[c]
            assert np.array_equal(a, b, equal_nan=True), (v, k)
