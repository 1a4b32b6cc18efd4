"""GPU parity tests of the round-2 extensions of the VEP drivers against the CPU oracle:
  * compute_ρg! / update_ρg! with T- and P-dependent densities (rheology/BuoyancyForces.jl:37-60,153-167; Stokes2D.jl:646,678; Stokes3D.jl:505,538)
  * strain softening of C and ϕ at EII_pl (rheology/StressUpdate.jl:305-381; the EII keyword of StressKernels.jl:1053-1105)
  * DisplacementBoundaryConditions (displacement2velocity! at the start, flow_bcs! on U; BoundaryConditions.jl:71-78)
  * the free_surface form of compute_V! / compute_Res! (VelocityKernels.jl:134-180,271-307)
  * the strain_increment variant of the 2D VEP driver (Stokes2D.jl:659-734; StressKernels.jl:1147-1302)
The GeoParams forms behind density and softening are ASSUMED (include/jrx.h): these tests pin HIP path == oracle, not the formulas.
Tolerance 1e-9 after tens of iterations (exp / erfc / sin of the device library differ from glibc in the last bits)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cp(a):
    return {k: v.copy(order="F") for k, v in a.items()}


def _phases_rho(phases, *, g=9.81):
    out = [dict(p) for p in phases]
    out[0].update(density=dict(kind="PT", rho0=3.3, alpha=2.0e-2, beta=1.0e-2, T0=0.5, P0=0.1), g=g)
    out[1].update(density=dict(kind="compressible", rho0=2.7, beta=3.0e-2, P0=0.0))
    return out


def _phases_soft(phases):
    out = [dict(p) for p in phases]
    out[0].update(softening_C=dict(kind="linear", min=0.4 * out[0]["C"], max=out[0]["C"], lo=0.02, hi=0.3),
                  softening_phi=dict(kind="nonlinear", xi0=30.0, Delta=10.0, mu=0.2, sigma=0.1))
    out[1].update(softening_C=dict(kind="nonlinear", xi0=out[1]["C"], Delta=0.5 * out[1]["C"], mu=0.1, sigma=0.05))
    return out


@pytest.mark.parametrize("ghosted", [False, True])
def test_vep2d_density_update_matches_oracle(jr, oracle, ghosted):
    """update_ρg! inside the multiphase solve!: args.T cell-centred (thermal.Tc), or -- ghosted -- thermal.T (ni .+ 2), which the reference reads at the cell's own
    [i, j] without a shift (getindex_NamedTuple(args, I...), BuoyancyForces.jl:52; test/test_sinking_block.jl:155,160 passes such an array)"""
    from justrelax_jl_amd.checks import max_rel_diff
    from justrelax_jl_amd.arrays import from_numpy
    from test_gpu_vep2d import _download, _upload, _vep_params
    s = jr.miniapps.shearband2d(40, iterMax=60, nout=20)
    s.kwargs.update(iterMin=10)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(5)
    s.arrays["T"] = np.asfortranarray(rng.uniform(0.0, 2.0, size=tuple(n + 2 for n in s.ni) if ghosted else s.ni))
    s.arrays["fy"][...] = 123.0                     # compute_ρg! must overwrite the caller's values
    phases = _phases_rho(s.extra["phases"], g=0.3)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(phases), _vep_params(oracle, s, iterMin=10, T_ghosted=ghosted))
    stokes, pr, ρg = _upload(jr, s)
    T = from_numpy(s.arrays["T"], stokes.P.device)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, dict(T=T, P=stokes.P), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 61
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    assert np.ptp(ref["fy"]) > 0 and not (ref["fy"] == 123.0).any()          # the density really varies and was recomputed
    assert max_rel_diff(jr.to_numpy(ρg[1]), ref["fy"]) <= 1e-12
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-9, k


def test_vep2d_softening_stress_update_matches_oracle(jr, oracle):
    """update_stresses_center_vertex_ps! with LinearSoftening / NonLinearSoftening laws and a non-trivial EII_pl field"""
    import torch
    from justrelax_jl_amd import _lib, stokes as st_mod
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _randomize, _upload, _vep_params
    s = jr.miniapps.shearband2d(24)
    _randomize(s)
    rng = np.random.default_rng(11)
    s.arrays["EII_pl"][...] = rng.uniform(0.0, 0.4, size=s.ni)
    phases = _phases_soft(s.extra["phases"])
    rh, p = oracle.rheology_struct(phases), _vep_params(oracle, s)
    ref = _cp(s.arrays)
    theta = np.asfortranarray(rng.uniform(-1, 1, size=s.ni))
    lam = np.asfortranarray(rng.uniform(0, 0.1, size=s.ni))
    lamv = np.asfortranarray(rng.uniform(0, 0.1, size=(s.ni[0] + 1, s.ni[1] + 1)))
    lam_r, lamv_r = lam.copy(order="F"), lamv.copy(order="F")
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    f = oracle.vep2d(ref)
    oracle.lib().orc_vep2d_stress(C.byref(f), dp(theta), dp(lam_r), dp(lamv_r), C.byref(rh), C.byref(p))
    # the same call without softening gives different stresses: the laws act
    ref0, lam0, lamv0 = _cp(s.arrays), lam.copy(order="F"), lamv.copy(order="F")
    f0 = oracle.vep2d(ref0)
    oracle.lib().orc_vep2d_stress(C.byref(f0), dp(theta), dp(lam0), dp(lamv0), C.byref(oracle.rheology_struct(s.extra["phases"])), C.byref(p))
    assert np.abs(ref["txx"] - ref0["txx"]).max() > 1e-3
    stokes, pr, ρg = _upload(jr, s)
    dev = stokes.P.device
    th_d, lam_d, lamv_d = from_numpy(theta, dev), from_numpy(lam, dev), from_numpy(lamv, dev)
    h = _lib.default_handle()
    fd = st_mod.vep_fields2d(stokes, ρg, pr)
    pd = st_mod.vep_params2d(stokes, s.pt, s.grid, s.flow_bcs, s.dt)
    h.call("jrx_vep2d_update_stresses", C.byref(fd), C.c_void_p(th_d.data_ptr()), C.c_void_p(lam_d.data_ptr()), C.c_void_p(lamv_d.data_ptr()),
           C.byref(st_mod.rheology_table(phases)), C.byref(pd))
    out = _download(jr, stokes)
    for k in ("txx", "tyy", "txy", "txy_c", "tII", "eta_vep", "P", "eplxx", "eplyy", "eplxy", "evol_pl"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-11, k
    assert max_rel_diff(jr.to_numpy(lam_d), lam_r) <= 1e-11 and max_rel_diff(jr.to_numpy(lamv_d), lamv_r) <= 1e-11


@pytest.mark.parametrize("free_surface", [False, True])
def test_vep2d_displacement_bcs_and_free_surface_match_oracle(jr, oracle, free_surface):
    from justrelax_jl_amd.arrays import DisplacementBoundaryConditions
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _upload, _vep_params
    s = jr.miniapps.shearband2d(24, iterMax=45, nout=15)
    s.kwargs.update(iterMin=10, free_surface=free_surface)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(2)
    s.arrays["fy"][...] = rng.uniform(-1.0, 1.0, size=s.ni)          # a density gradient for the free-surface term
    # the caller holds displacements: U = V dt; V itself starts as garbage and must be rebuilt by displacement2velocity!
    s.arrays["Ux"][...] = s.arrays["Vx"] * s.dt
    s.arrays["Uy"][...] = s.arrays["Vy"] * s.dt
    s.arrays["Vx"][...] = 7.0
    s.arrays["Vy"][...] = -3.0
    b = s.flow_bcs
    dbc = DisplacementBoundaryConditions(free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]),
                                      _vep_params(oracle, s, iterMin=10, free_surface=free_surface, displacement_bcs=True))
    stokes, pr, ρg = _upload(jr, s)
    r = jr.solve_(stokes, s.pt, s.grid, dbc, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 46
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-9, k


@pytest.mark.parametrize("ghosted", [False, True])
def test_vep3d_density_softening_displacement_match_oracle(jr, oracle, ghosted):
    from justrelax_jl_amd.arrays import DisplacementBoundaryConditions, from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep3d import _download, _params, _upload
    s = jr.miniapps.shearband3d((14, 10, 9), iterMax=24, nout=8)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(8)
    # args.T: cell-centred, or -- ghosted -- thermal.T (ni .+ 2) as the 3D miniapps pass it (RisingBlob3D/Blob3D.jl:355), read by update_ρg! at the cell's own index
    s.arrays["T"] = np.asfortranarray(rng.uniform(0.0, 2.0, size=tuple(n + 2 for n in s.ni) if ghosted else s.ni))
    s.arrays["EII_pl"][...] = rng.uniform(0.0, 0.4, size=s.ni)
    for k, v in (("Ux", "Vx"), ("Uy", "Vy"), ("Uz", "Vz")):
        s.arrays[k][...] = s.arrays[v] * s.dt
        s.arrays[v][...] = 1.0
    phases = _phases_soft(_phases_rho([dict(ph, Kb=3.0, psi_deg=5.0) for ph in s.extra["phases"]], g=0.2))
    b = s.flow_bcs
    dbc = DisplacementBoundaryConditions(free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes3d_vep_solve(ref, oracle.rheology_struct(phases), _params(oracle, s, displacement_bcs=True, T_ghosted=ghosted))
    stokes, pr, ρg = _upload(jr, s)
    T = from_numpy(s.arrays["T"], stokes.P.device)
    r = jr.solve_(stokes, s.pt, s.grid, dbc, ρg, pr, phases, dict(T=T), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 25
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    assert max_rel_diff(jr.to_numpy(ρg[2]), ref["fz"]) <= 1e-12 and np.ptp(ref["fz"]) > 0
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-9, k


def _nl_params(oracle, s, **over):
    pt, b = s.pt, s.flow_bcs
    kw = dict(iterMax=s.kwargs["iterMax"], nout=s.kwargs["nout"], stag_mode=1, cutoff=s.kwargs["viscosity_cutoff"], T_ghosted=True)
    kw.update(over)
    return oracle.vep_params2d(s.ni, s.grid._di["center"], s.dt, dict(r=pt.r, theta_dtau=pt.θ_dτ, eta_dtau=pt.ηdτ, eps_rel=pt.ϵ_rel, eps_abs=pt.ϵ_abs),
                               free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, **kw)


@pytest.mark.parametrize("plastic", [False, True])
def test_single_phase_driver_matches_oracle(jr, oracle, plastic):
    """solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology::MaterialParams, args, dt, igg) (Stokes2D.jl:345-557) on the Stokes problem of
    test/test_WENO5.jl (aspect ratio 1, so that the thermal anomaly is resolved): T-dependent Arrhenius viscosity with relaxation,
    PT_Density buoyancy, compute_τ_nonlinear! + center2vertex!; `plastic` adds the script's DruckerPrager_regularised (rheology_plastic)."""
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import VEP_MAP, _get
    import torch
    s = jr.miniapps.thermal_convection2d(32, ar=1, iterMax=299, nout=100)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    ph = dict(s.extra["rheology"])
    if plastic:      # the script's regularisation viscosity; cohesion chosen so that about half of the cells yield within the 300 iterations
        ph.update(C=8.0e6, phi_deg=0.0, psi_deg=0.0, eta_vp=1.0e16)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_nonlinear_solve(ref, oracle.rheology_struct([ph]), _nl_params(oracle, s))
    dev = torch.device("cuda", torch.cuda.current_device())
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP_MAP.items():
        _get(st, path).copy_(from_numpy(s.arrays[k], dev))
    ρg = (from_numpy(s.arrays["fx"], dev), from_numpy(s.arrays["fy"], dev))
    T = from_numpy(s.arrays["T"], dev)
    r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, ph, dict(T=T, P=st.P), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 300
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-8)
    if plastic:
        assert 100 < (ref["eplxx"] != 0).sum() < ref["eplxx"].size
    out = {k: jr.to_numpy(_get(st, path)) for k, path in VEP_MAP.items()}
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-8, k
    assert max_rel_diff(jr.to_numpy(ρg[1]), ref["fy"]) <= 1e-12


@pytest.mark.parametrize("nd", [3, 2])
def test_visco_elastic_drivers_with_displacement_bcs_match_oracle(jr, oracle, nd):
    """solve!(…, flow_bcs::DisplacementBoundaryConditions, …) of the visco-elastic drivers: displacement2velocity! first (Stokes3D.jl:72,
    Stokes2D.jl:223), flow_bcs! on @displacement afterwards (BoundaryConditions.jl:71-78) -- the ghosts of V are never refreshed."""
    from justrelax_jl_amd import checks
    from justrelax_jl_amd.arrays import DisplacementBoundaryConditions
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    s = jr.miniapps.random_fields3d((70, 9, 11), iterMax=20, nout=5) if nd == 3 else jr.miniapps.random_fields2d((33, 17), iterMax=20, nout=5)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    b = s.flow_bcs
    s.flow_bcs = DisplacementBoundaryConditions(free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic)
    for u, v in (("Ux", "Vx"), ("Uy", "Vy"), ("Uz", "Vz"))[:nd]:
        s.arrays[u][...] = s.arrays[v] * s.dt
        s.arrays[v][...] = 5.0
    ref = _cp(s.arrays)
    if nd == 3:
        r_ref = oracle.stokes3d_solve(ref, checks.oracle_params3d(oracle, s))
    else:
        r_ref = oracle.stokes2d_solve(ref, checks.oracle_params2d(oracle, s))
    stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    mats = (K, G) if nd == 3 else (G, K)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, *mats, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 21
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-10)
    d = checks.compare_stokes(download_stokes(stokes), ref)
    assert max(d.values()) <= 1e-9, d


@pytest.mark.parametrize("displacement,soft", [(False, False), (True, False), (True, True)])
def test_vep2d_strain_increment_matches_oracle(jr, oracle, displacement, soft):
    """strain_increment = true: ∇U / Δε from U = V dt every iteration, Δε form of update_stresses_center_vertex_ps!, flow_bcs! on U or V as the
    boundary-condition type says; yielding state (pre-stress near yield), dt not a power of two"""
    from justrelax_jl_amd.arrays import DisplacementBoundaryConditions
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _upload, _vep_params
    s = jr.miniapps.shearband2d(24, iterMax=45, nout=15)
    s.kwargs.update(iterMin=10, strain_increment=True)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    s.dt = 0.3
    rng = np.random.default_rng(8)
    for c in ("xx", "yy", "xy", "xy_c"):
        s.arrays["to" + c][...] = rng.uniform(-1.2, 1.2, size=s.arrays["to" + c].shape)
        s.arrays["t" + c][...] = s.arrays["to" + c]
    s.arrays["EII_pl"][...] = rng.uniform(0, 0.2, size=s.ni)
    s.arrays["Ux"][...] = s.arrays["Vx"] * s.dt
    s.arrays["Uy"][...] = s.arrays["Vy"] * s.dt
    for k in ("dexx", "deyy", "divU"):
        s.arrays[k] = np.zeros(s.ni, order="F")
    phases = _phases_soft(s.extra["phases"]) if soft else s.extra["phases"]
    b = s.flow_bcs
    bcs = DisplacementBoundaryConditions(free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic) if displacement else b
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(phases), _vep_params(oracle, s, iterMin=10, strain_increment=True, displacement_bcs=displacement))
    stokes, pr, ρg = _upload(jr, s)
    r = jr.solve_(stokes, s.pt, s.grid, bcs, ρg, pr, phases, None, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 46
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    assert (ref["eplxx"] != 0).any() and np.abs(ref["dexx"]).max() > 0
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-9, k
    for k, t in (("dexx", stokes.Δε.xx), ("deyy", stokes.Δε.yy), ("divU", getattr(stokes, "∇U"))):
        assert max_rel_diff(jr.to_numpy(t), ref[k]) <= 1e-9, k
    # and it is not the plain variant in disguise: the same call without the flag gives different stresses
    ref0 = _cp(s.arrays)
    oracle.stokes2d_vep_solve(ref0, oracle.rheology_struct(phases), _vep_params(oracle, s, iterMin=10, displacement_bcs=displacement))
    assert np.abs(ref0["txx"] - ref["txx"]).max() > 1e-12


def test_sinking_block_reference_test_on_the_device(jr, oracle):
    """test/test_sinking_block.jl:93-209 through the operator API as the script chains it: compute_ρg!(ρg[2], phase_ratios, rheology, args), init_P!,
    compute_viscosity!(stokes, phase_ratios, args, rheology, cutoff), flow_bcs!, solve!, compute_dt, velocity2vertex! -- err_evo1[end] < 1e-5, the velocity
    the reference's test prints (within 6 %), and every field equal to the oracle's solve"""
    import json
    from pathlib import Path
    import torch
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _upload, _vep_params
    ka = json.loads((Path(__file__).parent / "golden" / "reference_known_answers.json").read_text())["sinking_block2D"]
    s = jr.miniapps.sinking_block2d(32)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]), _vep_params(oracle, s, iterMin=100))
    # device: start from velocities / stresses only, derive ρg, P and η with the operators
    host_fy, host_P, host_eta = s.arrays["fy"].copy(), s.arrays["P"].copy(), s.arrays["eta"].copy()
    s.arrays["fy"][...] = 0.0
    s.arrays["P"][...] = 0.0
    s.arrays["eta"][...] = 1.0
    s.arrays["eta_v"][...] = 1.0
    st, pr, ρg = _upload(jr, s)
    dev = st.P.device
    args = dict(T=jr.fzeros(tuple(n + 2 for n in s.ni), dev, 1.0), P=st.P)
    jr.compute_ρg_(ρg, pr, s.extra["phases"], args)                                        # :155 (args.T ghosted, read unshifted; constant densities ignore it)
    assert np.array_equal(jr.to_numpy(ρg[1]), host_fy)
    st.P.copy_(ρg[1] * torch.tensor(np.abs(s.grid.xci[1]), device=dev)[None, :])          # init_P! :86-89
    jr.compute_viscosity_(st, pr, args, s.extra["phases"], (-np.inf, np.inf))              # :162
    assert np.allclose(jr.to_numpy(st.viscosity.η), host_eta, rtol=1e-14) and np.allclose(jr.to_numpy(st.P), host_P, rtol=1e-15)
    jr.flow_bcs_(st, s.flow_bcs)
    r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], args, s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] and r.err_evo1[-1] < ka["err_evo1_last_below"]
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-6)
    dt = jr.compute_dt_(st, s.extra["di"])
    assert dt == pytest.approx(0.9 * min(s.extra["di"][0] / np.abs(ref["Vx"]).max(), s.extra["di"][1] / np.abs(ref["Vy"]).max()), rel=1e-6)
    n = s.ni[0]
    Vx_v, Vy_v = jr.fzeros((n + 1, n + 1), dev), jr.fzeros((n + 1, n + 1), dev)
    jr.velocity2vertex_(Vx_v, Vy_v, st.V.Vx, st.V.Vy)
    vmax = float(torch.sqrt(Vx_v ** 2 + Vy_v ** 2).max())
    assert vmax == pytest.approx(ka["max_velocity"], rel=6e-2)
    out = _download(jr, st)
    for k in ("P", "Vx", "Vy", "txx", "tyy", "txy", "exx", "exy", "tII", "eta_vep"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-6, k


@pytest.mark.parametrize("rho", [False, True])
def test_vep2d_graph_replay_changes_nothing(jr, rho):
    """option loop_graphs: runs of unobserved iterations of the 2D visco-elasto-plastic loop replay as captured hipGraphs of 32 iterations (three launches each,
    the (τxx, τyy) sets ping-pong inside); every field, the iteration count and the error history equal those of plain launches -- yielding state, with and without
    the in-loop density update"""
    import ctypes as C
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.arrays import from_numpy
    from test_gpu_vep2d import _download, _upload
    h = _lib.default_handle(0)
    outs = []
    try:
        for g in (0, 1):
            h.call("jrx_set_option", C.c_char_p(b"loop_graphs"), C.c_int64(g))
            s = jr.miniapps.shearband2d(48, iterMax=299, nout=150)
            s.kwargs.update(iterMin=10)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
            rng = np.random.default_rng(8)
            for c in ("xx", "yy", "xy", "xy_c"):
                s.arrays["to" + c][...] = rng.uniform(-1.2, 1.2, size=s.arrays["to" + c].shape)
                s.arrays["t" + c][...] = s.arrays["to" + c]
            phases = _phases_rho(s.extra["phases"], g=0.3) if rho else s.extra["phases"]
            st, pr, ρg = _upload(jr, s)
            args = dict(T=from_numpy(np.asfortranarray(rng.uniform(0.0, 2.0, size=s.ni)), st.P.device), P=st.P) if rho else None
            r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, args, s.dt, None, kwargs=s.kwargs)
            outs.append((r.iter, list(r.err_evo1), _download(jr, st), jr.to_numpy(ρg[1])))
    finally:
        h.call("jrx_set_option", C.c_char_p(b"loop_graphs"), C.c_int64(1))
    a, b = outs
    assert a[0] == b[0] == 300 and a[1] == b[1]
    for k in a[2]:
        assert np.array_equal(a[2][k], b[2][k], equal_nan=True), k
    assert np.array_equal(a[3], b[3]) and (a[2]["eplxx"] != 0).any()


def test_single_phase_driver_graph_replay_changes_nothing(jr):
    """option loop_graphs in the single-phase non-linear driver: the replayed iteration (center2vertex! in one pass, flow_bcs! folded into compute_V!, 16
    iterations per graph) leaves every array, ghost entries included, and the error history as the plain launches do"""
    import ctypes as C
    import torch
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.arrays import from_numpy
    from test_gpu_vep2d import VEP_MAP, _get
    h = _lib.default_handle(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    outs = []
    try:
        for g in (0, 1):
            h.call("jrx_set_option", C.c_char_p(b"loop_graphs"), C.c_int64(g))
            s = jr.miniapps.thermal_convection2d(40, ar=1, iterMax=299, nout=100)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
            ph = dict(s.extra["rheology"], C=8.0e6, phi_deg=0.0, psi_deg=0.0, eta_vp=1.0e16)
            st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
            for k, path in VEP_MAP.items():
                _get(st, path).copy_(from_numpy(s.arrays[k], dev))
            ρg = (from_numpy(s.arrays["fx"], dev), from_numpy(s.arrays["fy"], dev))
            r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, ph, dict(T=from_numpy(s.arrays["T"], dev), P=st.P), s.dt, None, kwargs=s.kwargs)
            outs.append((r.iter, list(r.err_evo1), {k: jr.to_numpy(_get(st, path)) for k, path in VEP_MAP.items()}, jr.to_numpy(ρg[1])))
    finally:
        h.call("jrx_set_option", C.c_char_p(b"loop_graphs"), C.c_int64(1))
    a, b = outs
    assert a[0] == b[0] == 300 and a[1] == b[1]
    for k in a[2]:
        assert np.array_equal(a[2][k], b[2][k], equal_nan=True), k
    assert np.array_equal(a[3], b[3])
