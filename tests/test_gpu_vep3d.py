"""GPU parity tests, 3D multiphase visco-elasto-plastic Stokes (Stokes3D.jl:447-668) vs the CPU oracle.
Tolerance 1e-12 of each field's max for one kernel call (observed: bit-identical), 1e-9 after tens of PT iterations."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_T = {"e": "ε", "epl": "ε_pl", "de": "Δε", "t": "τ", "to": "τ_o"}
VEP3_MAP = dict(P="P", P0="P0", divV="divV", Q="Q", Vx="V.Vx", Vy="V.Vy", Vz="V.Vz", Ux="U.Ux", Uy="U.Uy", Uz="U.Uz", tII="τ.II",
                eta="viscosity.η", eta_vep="viscosity.η_vep", EII_pl="EII_pl", evol_pl="ε_vol_pl", EVol_pl="EVol_pl",
                RP="R.RP", Rx="R.Rx", Ry="R.Ry", Rz="R.Rz", omega_yz="ω.yz", omega_xz="ω.xz", omega_xy="ω.xy")
for _pre, _t in _T.items():
    for _c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):
        if _pre == "de" and _c in ("xx", "yy", "zz"):
            continue
        VEP3_MAP[_pre + _c] = f"{_t}.{_c}"


def _get(o, path):
    for p in path.split("."):
        o = getattr(o, p)
    return o


def _upload(jr, s):
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP3_MAP.items():
        _get(st, path).copy_(from_numpy(s.arrays[k], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    for k, name in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
        getattr(pr, name).copy_(from_numpy(s.arrays[k], dev))
    ρg = tuple(from_numpy(s.arrays[k], dev) for k in ("fx", "fy", "fz"))
    return st, pr, ρg


def _download(jr, st):
    return {k: jr.to_numpy(_get(st, path)) for k, path in VEP3_MAP.items()}


def _params(orc, s, **over):
    pt, b = s.pt, s.flow_bcs
    kw = dict(iterMax=s.kwargs["iterMax"], nout=s.kwargs["nout"])
    kw.update(over)
    return orc.vep_params3d(s.ni, s.grid._di["center"], s.dt, dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                            free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, **kw)


def _randomize(s, seed=4, Kb=3.0, psi=6.0):
    rng = np.random.default_rng(seed)
    a = s.arrays
    for pre in ("e", "t", "to"):
        for c in ("xx", "yy", "zz", "yz", "xz", "xy"):
            a[pre + c][...] = rng.uniform(-2.0, 2.0, size=a[pre + c].shape)
    for k in ("P", "tyz_c", "txz_c", "txy_c", "toyz_c", "toxz_c", "toxy_c"):
        a[k][...] = rng.uniform(-2.0, 2.0, size=a[k].shape)
    a["eta"][...] = 10.0 ** rng.uniform(-1.0, 0.5, size=a["eta"].shape)
    for k in ("phase_c", "phase_yz", "phase_xz", "phase_xy"):
        r = rng.uniform(0.0, 1.0, size=a[k].shape[1:])
        r[rng.uniform(size=r.shape) < 0.3] = 0.0
        r[rng.uniform(size=r.shape) < 0.3] = 1.0
        a[k][0], a[k][1] = r, 1.0 - r
    return [dict(ph, Kb=Kb, psi_deg=psi) for ph in s.extra["phases"]]


@pytest.mark.parametrize("edges,ni", [(0, (13, 9, 7)), (0, (70, 5, 6)), (1, (13, 9, 7)), (1, (70, 5, 6)), (1, (125, 6, 18)), (1, (61, 7, 33)),
                                      (1, (62, 4, 16)), (1, (3, 3, 3)), (2, (70, 5, 6)), (2, (13, 9, 7)),
                                      (3, (13, 9, 7)), (3, (70, 5, 6)), (3, (125, 6, 18)), (3, (61, 7, 33)), (3, (62, 4, 16)), (3, (3, 3, 3)),
                                      (4, (13, 9, 7)), (4, (70, 5, 6)), (4, (125, 6, 18)), (4, (61, 7, 33)), (4, (62, 4, 16)), (4, (3, 3, 3)),
                                      (6, (13, 9, 7)), (6, (70, 5, 6)), (6, (125, 6, 18)), (6, (61, 7, 33)), (6, (62, 4, 16)), (6, (3, 3, 3))])
def test_update_stresses_3d_matches_oracle(jr, oracle, edges, ni):
    """update_stresses_center_vertex_ps! 3D (StressKernels.jl:671-989) on random states: yielding and elastic nodes, mixed
    phase ratios, dilatant plasticity with finite bulk modulus.  edges = 1: the z-marching edge kernel, one family per block (grids spanning one, two and three
    62-node lane segments and one to three z chunks), 2: the same as one launch per family, 3: the three family waves of a row as one workgroup sharing the
    centre operands through LDS, 4 (default): the shear operands too, 0: one node per thread."""
    from justrelax_jl_amd import _lib, stokes as st_mod
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.shearband3d(ni)
    phases = _randomize(s)
    rh = oracle.rheology_struct(phases)
    p = _params(oracle, s)
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    rng = np.random.default_rng(9)
    theta = np.asfortranarray(rng.uniform(-1, 1, size=s.ni))
    lam = np.asfortranarray(rng.uniform(0, 0.1, size=s.ni))
    lamv = [np.asfortranarray(rng.uniform(0, 0.1, size=s.arrays[k].shape)) for k in ("tyz", "txz", "txy")]
    lam_r, lamv_r = lam.copy(order="F"), [x.copy(order="F") for x in lamv]
    oracle.vep3d_stress(ref, theta, lam_r, lamv_r, rh, p)
    stokes, pr, ρg = _upload(jr, s)
    dev = stokes.P.device
    th_d, lam_d = from_numpy(theta, dev), from_numpy(lam, dev)
    lamv_d = [from_numpy(x, dev) for x in lamv]
    h = _lib.default_handle()
    fd = st_mod.vep_fields3d(stokes, ρg, pr)
    pd = st_mod.vep_params3d(stokes, s.pt, s.grid, s.flow_bcs, s.dt)
    lv = (C.c_void_p * 3)(*[x.data_ptr() for x in lamv_d])
    h.call("jrx_tuning_set", C.c_char_p(b"vep3_edges"), C.c_int64(edges))
    try:
        h.call("jrx_vep3d_update_stresses", C.byref(fd), C.c_void_p(th_d.data_ptr()), C.c_void_p(lam_d.data_ptr()), lv,
               C.byref(st_mod.rheology_table(phases)), C.byref(pd))
    finally:
        h.call("jrx_tuning_set", C.c_char_p(b"vep3_edges"), C.c_int64(4))
    out = _download(jr, stokes)
    assert (lam_r != lam).any() and (lam_r == lam).any() and (ref["eplxz"] != 0).any() and (ref["eplxz"] == 0).any()
    for k in ("txx", "tyy", "tzz", "tyz", "txz", "txy", "tyz_c", "txz_c", "txy_c", "tII", "eta_vep", "P", "eplxx", "eplyy", "eplzz", "eplyz", "eplxz",
              "eplxy", "evol_pl"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-12, k
    assert max_rel_diff(jr.to_numpy(lam_d), lam_r) <= 1e-12
    for a, b in zip(lamv_d, lamv_r):
        assert max_rel_diff(jr.to_numpy(a), b) <= 1e-12


def test_vep3d_solve_matches_oracle_over_iterations(jr, oracle):
    """the whole driver (pressure with phase ratios, viscosity relaxation, stress kernel, velocity sweep, BCs, norms, epilogue)
    for a fixed number of iterations on a yielding state"""
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.shearband3d((20, 12, 10), iterMax=39, nout=10)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(3)
    for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):     # pre-stress close to yield so that plasticity is active
        s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
        s.arrays["t" + c][...] = s.arrays["to" + c]
    s.arrays["EII_pl"][...] = rng.uniform(0, 0.1, size=s.ni)
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r_ref = oracle.stokes3d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]), _params(oracle, s))
    stokes, pr, ρg = _upload(jr, s)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    out = _download(jr, stokes)
    assert r.iter == r_ref["iter"] == 40
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9) and np.allclose(r.norm_Rz, r_ref["norm_Rz"], rtol=1e-9)
    assert (ref["eplxx"] != 0).any()
    from justrelax_jl_amd.checks import interior_mask3d
    # (∇V, RP, ε_pl, ε_vol_pl, η_vep are stored by observed iterations only -- the last one is: they must still be the reference's)
    for k in ("P", "P0", "Vx", "Vy", "Vz", "Ux", "txx", "tyy", "tzz", "tyz", "txz", "txy", "tyz_c", "tII", "eta", "eta_vep", "exx", "exz", "eplxx", "eplyy", "eplzz",
              "eplxy", "eplxz", "eplyz", "evol_pl", "divV", "EII_pl", "EVol_pl", "Rx", "Rz", "RP", "toxx", "toxz", "toxy_c", "omega_yz", "omega_xz", "omega_xy", "exz_c", "eplyz_c"):
        m = interior_mask3d(k, ref[k].shape)
        scale = max(np.abs(ref[k]).max(), 1e-300)
        assert np.abs(out[k] - ref[k])[m].max() <= 1e-9 * scale, k


@pytest.mark.parametrize("nphase", [1, 3, 4, 5])
def test_update_stresses_3d_other_phase_counts(jr, oracle, nphase):
    """the z-marching edge kernel is instantiated per phase count (1..4, unrolled phase loops); five phases take the one-node-per-thread kernel"""
    from justrelax_jl_amd import _lib, stokes as st_mod
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    import torch
    ni = (66, 7, 19)
    s = jr.miniapps.shearband3d(ni)
    _randomize(s)
    rng = np.random.default_rng(40 + nphase)
    base = s.extra["phases"]
    phases = []
    for q in range(nphase):
        ph = dict(base[q % 2], Kb=3.0 + 0.5 * q, psi_deg=4.0 + q)
        ph["eta"] = ph["eta"] * (1.0 + 0.3 * q)
        ph["C"] = ph["C"] * (1.0 - 0.1 * q)
        ph["G"] = ph["G"] * (1.0 + 0.2 * q)
        phases.append(ph)
    for k in ("phase_c", "phase_yz", "phase_xz", "phase_xy"):
        shp = s.arrays[k].shape[1:]
        r = rng.dirichlet(np.ones(nphase), size=shp)                      # ratios sum to 1
        r[rng.uniform(size=shp) < 0.3] = np.eye(nphase)[rng.integers(nphase)]    # some pure cells
        s.arrays[k] = np.asfortranarray(np.moveaxis(r, -1, 0))
    rh = oracle.rheology_struct(phases)
    p = _params(oracle, s)
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    theta = np.asfortranarray(rng.uniform(-1, 1, size=s.ni))
    lam = np.asfortranarray(rng.uniform(0, 0.1, size=s.ni))
    lamv = [np.asfortranarray(rng.uniform(0, 0.1, size=s.arrays[k].shape)) for k in ("tyz", "txz", "txy")]
    lam_r, lamv_r = lam.copy(order="F"), [x.copy(order="F") for x in lamv]
    oracle.vep3d_stress(ref, theta, lam_r, lamv_r, rh, p)
    dev = torch.device("cuda", torch.cuda.current_device())
    stokes = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP3_MAP.items():
        _get(stokes, path).copy_(from_numpy(s.arrays[k], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, nphase, s.ni)
    for k, name in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
        getattr(pr, name).copy_(from_numpy(s.arrays[k], dev))
    ρg = tuple(from_numpy(s.arrays[k], dev) for k in ("fx", "fy", "fz"))
    th_d, lam_d = from_numpy(theta, dev), from_numpy(lam, dev)
    lamv_d = [from_numpy(x, dev) for x in lamv]
    h = _lib.default_handle()
    fd = st_mod.vep_fields3d(stokes, ρg, pr)
    pd = st_mod.vep_params3d(stokes, s.pt, s.grid, s.flow_bcs, s.dt)
    lv = (C.c_void_p * 3)(*[x.data_ptr() for x in lamv_d])
    h.call("jrx_vep3d_update_stresses", C.byref(fd), C.c_void_p(th_d.data_ptr()), C.c_void_p(lam_d.data_ptr()), lv, C.byref(st_mod.rheology_table(phases)), C.byref(pd))
    out = _download(jr, stokes)
    assert (ref["eplxz"] != 0).any() and (ref["eplxz"] == 0).any()
    for k in ("txx", "tyy", "tzz", "tyz", "txz", "txy", "tyz_c", "txz_c", "txy_c", "tII", "eta_vep", "P", "eplxx", "eplyz", "eplxz", "eplxy", "evol_pl"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-12, k
    for a, b in zip(lamv_d, lamv_r):
        assert max_rel_diff(jr.to_numpy(a), b) <= 1e-12


@pytest.mark.parametrize("ni,iters", [((96, 80, 72), 12), ((160, 160, 160), 6), ((256, 256, 256), 4)])
def test_vep3d_solve_matches_oracle_on_a_multi_tile_grid(jr, oracle, ni, iters):
    """the whole 3D VEP driver on grids that span several 62-node lane segments, row blocks, 16-plane chunks of the z-marching edge kernel and z chunks of
    the pre kernel (160^3 and 256^3: more tiles than fit on the chip at once, every XCD slab populated; 256^3 is the size the bench quotes), yielding state, two checks"""
    from justrelax_jl_amd.checks import interior_mask3d
    s = jr.miniapps.shearband3d(ni, iterMax=iters - 1, nout=iters // 2)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(13)
    for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):
        s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
        s.arrays["t" + c][...] = s.arrays["to" + c]
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r_ref = oracle.stokes3d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]), _params(oracle, s))
    stokes, pr, ρg = _upload(jr, s)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    out = _download(jr, stokes)
    assert r.iter == r_ref["iter"] == iters
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    assert (ref["eplyz"] != 0).any() and (ref["eplyz"] == 0).any()
    for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "tyz_c", "txz_c", "txy_c", "tII", "eta", "eta_vep", "eplxx", "eplyz", "eplxz",
              "eplxy", "EII_pl", "EVol_pl", "Rx", "Ry", "Rz", "RP"):
        m = interior_mask3d(k, ref[k].shape)
        scale = max(np.abs(ref[k]).max(), 1e-300)
        assert np.abs(out[k] - ref[k])[m].max() <= 1e-9 * scale, k


def test_tensor_invariant_and_viscosity_3d(jr, oracle):
    import torch
    from justrelax_jl_amd import stokes as st_mod
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.shearband3d((9, 8, 7))
    phases = _randomize(s, seed=7)
    stokes, pr, ρg = _upload(jr, s)
    st_mod.tensor_invariant_(stokes.τ)
    want = np.zeros(s.ni, order="F")
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    a = s.arrays
    oracle.lib().orc_tensor_invariant3d(dp(want), *[dp(a[k]) for k in ("txx", "tyy", "tzz", "tyz", "txz", "txy")], *[C.c_int64(n) for n in s.ni])
    assert max_rel_diff(jr.to_numpy(stokes.τ.II), want) <= 1e-14
    st_mod.compute_viscosity_(stokes, pr, None, phases, relaxation=0.3)
    ref = {k: v.copy(order="F") for k, v in a.items()}
    f = oracle.vep3d(ref)
    oracle.lib().orc_compute_viscosity3d(C.byref(f), C.byref(oracle.rheology_struct(phases)), C.byref(_params(oracle, s)), C.c_double(0.3))
    assert max_rel_diff(jr.to_numpy(stokes.viscosity.η), ref["eta"]) <= 1e-14


def test_epilogue_operators_3d(jr, oracle):
    """shear2center!, accumulate_tensor!, accumulate_vol!, compute_vorticity! 3D as stand-alone operators, bit-exact"""
    from justrelax_jl_amd import stokes as st_mod
    s = jr.miniapps.shearband3d((11, 7, 6))
    _randomize(s, seed=2)
    rng = np.random.default_rng(8)
    a = s.arrays
    for k in ("Vx", "Vy", "Vz", "eplxx", "eplyy", "eplzz", "eplyz", "eplxz", "eplxy", "EII_pl", "EVol_pl", "evol_pl"):
        a[k][...] = rng.uniform(-1, 1, size=a[k].shape)
    stokes, pr, ρg = _upload(jr, s)
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    n = [C.c_int64(m) for m in s.ni]
    L = oracle.lib()
    # shear2center!
    st_mod.shear2center_(stokes.ε)
    want = [np.zeros(s.ni, order="F") for _ in range(3)]
    L.orc_shear2center3d(*[dp(w) for w in want], dp(a["eyz"]), dp(a["exz"]), dp(a["exy"]), *n)
    for w, k in zip(want, ("yz_c", "xz_c", "xy_c")):
        assert np.array_equal(jr.to_numpy(getattr(stokes.ε, k)), w), k
    # accumulate_tensor! / accumulate_vol!
    II = np.zeros(s.ni, order="F")
    L.orc_tensor_invariant3d(dp(II), *[dp(a[k]) for k in ("eplxx", "eplyy", "eplzz", "eplyz", "eplxz", "eplxy")], *n)
    st_mod.accumulate_tensor_(stokes.EII_pl, stokes.ε_pl, 0.37)
    st_mod.accumulate_vol_(stokes.EVol_pl, stokes.ε_vol_pl, 0.37)
    assert np.array_equal(jr.to_numpy(stokes.EII_pl), a["EII_pl"] + II * 0.37)
    assert np.array_equal(jr.to_numpy(stokes.EVol_pl), a["EVol_pl"] + 0.37 * a["evol_pl"])
    # compute_vorticity!
    st_mod.compute_vorticity_(stokes, s.grid)
    w = [np.zeros(a[k].shape, order="F") for k in ("tyz", "txz", "txy")]
    _di = s.grid._di["center"]
    L.orc_compute_vorticity3d(*[dp(x) for x in w], dp(a["Vx"]), dp(a["Vy"]), dp(a["Vz"]), *n, *[C.c_double(d) for d in _di])
    for x, k in zip(w, ("yz", "xz", "xy")):
        assert np.array_equal(jr.to_numpy(getattr(stokes.ω, k)), x), k


@pytest.mark.parametrize("tile", [0, 1])
@pytest.mark.parametrize("ni,iters,nout,prekz", [((20, 12, 10), 7, 3, 0), ((20, 12, 10), 40, 10, 0), ((13, 9, 7), 5, 2, 1), ((70, 5, 6), 6, 5, 4), ((64, 64, 17), 4, 2, 8),
                                                   ((63, 31, 16), 5, 4, 16), ((127, 66, 40), 6, 3, 8), ((127, 66, 40), 5, 10 ** 9, 32), ((96, 80, 72), 35, 17, 0)])
def test_vep3d_fused_pre_centre_equals_the_three_kernels(jr, ni, iters, nout, prekz, tile):
    """k_vep3_prec (pre + viscosity relaxation + centre pass in one kernel ahead of the edge pass, second sets of η and τxx/τyy/τzz adopted by pointer swap; the default without
    neighbours) against the three kernels with the centre pass behind the edge pass: every field of the solve bit for bit -- odd and even iteration counts (the copy-back of the
    second sets), observed and unobserved iterations (the output-only arrays), chunk depths that cut the column at every place, a run long enough for the captured graphs"""
    from justrelax_jl_amd import _lib
    h = _lib.default_handle()
    outs, res = [], []
    try:
        for fuse in (0, 1):
            h.set_option("vep3_fuse_pc", fuse)
            h.set_option("vep3_prekz", prekz)
            h.set_option("vep3_prec_tile", tile)
            s = jr.miniapps.shearband3d(ni, iterMax=iters - 1, nout=nout)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            rng = np.random.default_rng(3)
            for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):     # pre-stress close to yield so that plasticity is active
                s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
                s.arrays["t" + c][...] = s.arrays["to" + c]
            stokes, pr, ρg = _upload(jr, s)
            n0 = h.get_option("stat_vep3_fused")
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
            assert r.iter == iters
            assert (h.get_option("stat_vep3_fused") > n0) == bool(fuse)
            outs.append(_download(jr, stokes))
            res.append(r)
    finally:
        h.set_option("vep3_fuse_pc", 1)
        h.set_option("vep3_prekz", 0)
        h.set_option("vep3_prec_tile", 2)
    assert (outs[0]["eplxx"] != 0).any() or iters < 5
    assert np.array_equal(np.asarray(res[0].err_evo1), np.asarray(res[1].err_evo1))
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("forces", ["none", "z", "z and one denormal in x", "xyz"])
def test_vep3d_velocity_sweep_does_not_load_body_forces_that_are_zero(jr, forces):
    """ShearBand3D.jl:114 hands three ρg arrays of zeros, a model with gravity has them in x and y.  The 3D VEP driver looks at their bits once per solve and the z-marching velocity
    sweep of unobserved iterations does not load the arrays that hold only +0.0 (k_velocity3d_zb, NOF; grids above 88^3 cells run that kernel) -- every field equals the run that
    loads them (tuning switch zero_forces = 0), bit for bit."""
    from justrelax_jl_amd import _lib
    h = _lib.default_handle()
    ni = (112, 88, 72)
    outs, res = [], []
    try:
        for zf in (1, 0):
            h.set_option("zero_forces", zf)
            s = jr.miniapps.shearband3d(ni, iterMax=6, nout=3)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            rng = np.random.default_rng(8)
            for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):
                s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
                s.arrays["t" + c][...] = s.arrays["to" + c]
            if "z" in forces:
                s.arrays["fz"][...] = rng.uniform(-1.0, 1.0, size=s.arrays["fz"].shape)
            if forces == "xyz":
                s.arrays["fx"][...] = rng.uniform(-1.0, 1.0, size=s.arrays["fx"].shape)
                s.arrays["fy"][...] = rng.uniform(-1.0, 1.0, size=s.arrays["fy"].shape)
            if "denormal" in forces:
                s.arrays["fx"][5, 7, 9] = 5e-324
            stokes, pr, ρg = _upload(jr, s)
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
            assert r.iter == 7
            outs.append(_download(jr, stokes))
            res.append(r)
    finally:
        h.set_option("zero_forces", 1)
    assert np.array_equal(np.asarray(res[0].err_evo1), np.asarray(res[1].err_evo1))
    for k in outs[0]:
        a, b = np.ascontiguousarray(outs[0][k]), np.ascontiguousarray(outs[1][k])
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) or (np.array_equal(a, b) and k[0] == "e"), k


@pytest.mark.parametrize("nphase", [1, 3, 4, 5])
def test_vep3d_solve_with_other_phase_counts_fused_equals_unfused(jr, nphase):
    """the whole 3D VEP solve with 1, 3, 4 (template instantiations of the fused pre / centre, centre and per-node edge kernels) and 5 phases (run-time phase loops everywhere): the
    fused kernel against the three kernels, and the constant-phase-count instantiations against the run-time loops, every field bit for bit"""
    import torch
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.arrays import from_numpy
    ni = (66, 9, 19)
    h = _lib.default_handle()
    outs = []
    try:
        for fuse, npc in ((0, 0), (1, 1), (1, 0), (0, 1)):
            h.set_option("vep3_fuse_pc", fuse)
            h.set_option("vep3_np_const", npc)
            s = jr.miniapps.shearband3d(ni, iterMax=11, nout=5)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            rng = np.random.default_rng(40 + nphase)
            for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):
                s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
                s.arrays["t" + c][...] = s.arrays["to" + c]
            base = s.extra["phases"]
            phases = []
            for q in range(nphase):
                ph = dict(base[q % 2], Kb=3.0 + 0.5 * q, psi_deg=4.0 + q)
                ph["eta"] = ph["eta"] * (1.0 + 0.3 * q)
                ph["C"] = ph["C"] * (1.0 - 0.1 * q)
                ph["G"] = ph["G"] * (1.0 + 0.2 * q)
                phases.append(ph)
            for k in ("phase_c", "phase_yz", "phase_xz", "phase_xy"):
                shp = s.arrays[k].shape[1:]
                r = rng.dirichlet(np.ones(nphase), size=shp)
                r[rng.uniform(size=shp) < 0.3] = np.eye(nphase)[rng.integers(nphase)]
                s.arrays[k] = np.asfortranarray(np.moveaxis(r, -1, 0))
            dev = torch.device("cuda", torch.cuda.current_device())
            stokes = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
            for k, path in VEP3_MAP.items():
                _get(stokes, path).copy_(from_numpy(s.arrays[k], dev))
            pr = jr.PhaseRatios(jr.AMDGPUBackend, nphase, s.ni)
            for k, name in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
                getattr(pr, name).copy_(from_numpy(s.arrays[k], dev))
            ρg = tuple(from_numpy(s.arrays[k], dev) for k in ("fx", "fy", "fz"))
            r_ = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, None, s.dt, None, kwargs=s.kwargs)
            assert r_.iter == 12
            outs.append(_download(jr, stokes))
    finally:
        h.set_option("vep3_fuse_pc", 1)
        h.set_option("vep3_np_const", 1)
    assert (outs[0]["eplxx"] != 0).any()
    for v in (1, 2, 3):
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[v][k], equal_nan=True), (v, k)
