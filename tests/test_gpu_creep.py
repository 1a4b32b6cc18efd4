"""GPU parity of the power-law creep (visc_kind 2: GeoParams DislocationCreep, r = 0; forms ASSUMED, see include/jrx.h) against the CPU oracle: the viscosity
kernels in both forms (compute_viscosity! from the strain rate, update_viscosity_τII! from the stress; rheology/Viscosity.jl:382-418,455-503,169-196) and the three
drivers that call them every iteration (Stokes2D.jl:345-557 single material, :577-866 multiphase, Stokes3D.jl:447-668).  Tolerances: 1e-12 for one kernel
evaluation (pow / exp of the device library against glibc), 1e-8 after tens of iterations of a solve whose viscosity depends on the stress it produces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CREEP = [dict(kind="dislocation", A=0.5, n=3.0, E=1.0, V=0.1, R=1.0), dict(kind="dislocation", A=2.0, n=3.3, E=0.6, V=0.0, R=1.0, apparatus="Invariant")]


def _cp(a):
    return {k: v.copy(order="F") for k, v in a.items()}


def _creep_phases(base):
    return [dict({k: v for k, v in ph.items() if k != "eta"}, creep=CREEP[q % 2]) for q, ph in enumerate(base)]      # no LinearViscous element beside the creep


def _ratios(rng, shape):
    r = rng.uniform(0, 1, size=shape)
    r[rng.uniform(size=shape) < 0.3] = 1.0
    r[rng.uniform(size=shape) < 0.1] = 0.0
    return np.asfortranarray(np.stack([r, 1.0 - r]))


@pytest.mark.parametrize("ghosted", [False, True])
@pytest.mark.parametrize("tau", [False, True])
def test_phase_viscosity_2d_matches_oracle(jr, oracle, tau, ghosted):
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _upload
    s = jr.miniapps.shearband2d(37, iterMax=1, nout=1)
    nx, ny = s.ni
    rng = np.random.default_rng(21)
    a = s.arrays
    for k in ("txx", "tyy", "txy_c", "exx", "eyy", "exy_c", "txy", "exy", "P"):
        a[k][...] = rng.uniform(-1, 1, size=a[k].shape)
    for k in ("txx", "tyy", "txy_c", "exx", "eyy", "exy_c"):
        a[k][2, 3] = 0.0
    a["phase_c"][...] = _ratios(rng, s.ni); a["phase_v"][...] = _ratios(rng, (nx + 1, ny + 1))
    a["eta"][...] = 1.0; a["eta_v"] = np.asfortranarray(np.full((nx + 1, ny + 1), 2.0))
    a["T"] = np.asfortranarray(rng.uniform(0.8, 1.5, size=(nx + 2, ny + 2) if ghosted else s.ni))
    phases = _creep_phases(s.extra["phases"])
    stokes, pr, _ = _upload(jr, s)
    stokes.viscosity.ηv.copy_(from_numpy(a["eta_v"], stokes.P.device))
    T = from_numpy(a["T"], stokes.P.device)
    p = oracle.vep_params2d(s.ni, s.grid._di["center"], s.dt, dict(r=0.7, theta_dtau=1.0, eta_dtau=1.0, eps_rel=0, eps_abs=0), cutoff=(1e-3, 1e3), T_ghosted=ghosted)
    oracle.compute_viscosity2d(a, oracle.rheology_struct(phases), p, nu=0.3, tau=tau)
    fn = jr.compute_viscosity_τII_ if tau else jr.compute_viscosity_
    fn(stokes, pr, dict(T=T, P=stokes.P), phases, (1e-3, 1e3), relaxation=0.3)
    assert np.ptp(a["eta"]) > 0.1 and np.isfinite(a["eta"]).all()
    assert max_rel_diff(jr.to_numpy(stokes.viscosity.η), a["eta"]) <= 1e-12
    assert max_rel_diff(jr.to_numpy(stokes.viscosity.ηv), a["eta_v"]) <= 1e-12


@pytest.mark.parametrize("tau", [False, True])
def test_phase_viscosity_3d_and_array_form_match_oracle(jr, oracle, tau):
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep3d import _upload
    s = jr.miniapps.shearband3d((13, 9, 7), iterMax=1, nout=1)
    nx, ny, nz = s.ni
    rng = np.random.default_rng(22)
    a = s.arrays
    for pre in "te":
        for k in ("xx", "yy", "zz", "yz", "xz", "xy"):
            a[pre + k][...] = rng.uniform(-1, 1, size=a[pre + k].shape)
        for k in ("xx", "yy", "zz"):
            a[pre + k][1, 2, 3] = 0.0
    a["P"][...] = rng.uniform(-1, 1, size=s.ni)
    a["phase_c"][...] = _ratios(rng, s.ni); a["eta"][...] = 1.0
    a["T"] = np.asfortranarray(rng.uniform(0.8, 1.5, size=(nx + 2, ny + 2, nz + 2)))
    phases = _creep_phases(s.extra["phases"])
    stokes, pr, _ = _upload(jr, s)
    T = from_numpy(a["T"], stokes.P.device)
    p = oracle.vep_params3d(s.ni, s.grid._di["center"], s.dt, dict(r=0.7, theta_dtau=1.0, eta_dtau=1.0, eps_rel=0, eps_abs=0), cutoff=(1e-3, 1e3), T_ghosted=True)
    oracle.compute_viscosity3d(a, oracle.rheology_struct(phases), p, nu=0.6, tau=tau)
    fn = jr.compute_viscosity_τII_ if tau else jr.compute_viscosity_
    fn(stokes, pr, dict(T=T, P=stokes.P), phases, (1e-3, 1e3), relaxation=0.6)
    assert np.ptp(a["eta"]) > 0.1
    assert max_rel_diff(jr.to_numpy(stokes.viscosity.η), a["eta"]) <= 1e-12
    # array form: compute_viscosity_εII! / _τII!(η, ν, AII, args, rheology, cutoff) for one MaterialParams
    AII = np.asfortranarray(rng.uniform(0.05, 3.0, size=s.ni)); eta = np.asfortranarray(np.full(s.ni, 0.5))
    one = dict(phases[0])
    oracle.compute_viscosity_single(eta, oracle.rheology_struct([one]), a["T"], a["P"], cutoff=(1e-3, 1e3), nu=0.5, AII=AII, tau=tau)
    stokes.viscosity.η.fill_(0.5)
    jr.compute_viscosity_(stokes, dict(T=T, P=stokes.P), one, (1e-3, 1e3), relaxation=0.5, fn="τII" if tau else "εII", AII=from_numpy(AII, stokes.P.device))
    assert max_rel_diff(jr.to_numpy(stokes.viscosity.η), eta) <= 1e-12
    with pytest.raises(Exception, match="invariant"):
        jr.compute_viscosity_(stokes, dict(T=T, P=stokes.P), one, (1e-3, 1e3))


def test_vep2d_solve_with_power_law_creep_matches_oracle(jr, oracle):
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _upload, _vep_params
    s = jr.miniapps.shearband2d(40, iterMax=60, nout=20)
    s.kwargs.update(iterMin=10, viscosity_cutoff=(1e-2, 1e2), viscosity_relaxation=0.1)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(23)
    nx, ny = s.ni
    s.arrays["T"] = np.asfortranarray(rng.uniform(0.9, 1.3, size=(nx + 2, ny + 2)))
    s.arrays["eta_v"] = np.asfortranarray(np.ones((nx + 1, ny + 1)))
    phases = _creep_phases(s.extra["phases"])
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(phases), _vep_params(oracle, s, iterMin=10, T_ghosted=True, cutoff=(1e-2, 1e2), viscosity_relaxation=0.1))
    stokes, pr, ρg = _upload(jr, s)
    stokes.viscosity.ηv.fill_(1.0)
    T = from_numpy(s.arrays["T"], stokes.P.device)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, dict(T=T, P=stokes.P), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 61
    assert max_rel_diff(jr.to_numpy(stokes.viscosity.ηv), ref["eta_v"]) <= 1e-8
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-8)
    assert np.ptp(ref["eta"]) > 1e-3                       # the viscosity really followed the stress
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-8, k


def test_vep3d_solve_with_power_law_creep_matches_oracle_and_heats(jr, oracle):
    """the chain of test/test_shearheating3D.jl:137-162 on the device: solve! (dislocation-creep phases) -> compute_shear_heating!, whose result the reference's
    test requires to be non-negative (:251)"""
    from types import SimpleNamespace
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep3d import _download, _params, _upload
    s = jr.miniapps.shearband3d((14, 10, 9), iterMax=24, nout=8)
    s.kwargs.update(viscosity_cutoff=(1e-2, 1e2), viscosity_relaxation=0.1)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(24)
    s.arrays["T"] = np.asfortranarray(rng.uniform(0.9, 1.3, size=tuple(n + 2 for n in s.ni)))
    phases = _creep_phases(s.extra["phases"])
    ref = _cp(s.arrays)
    r_ref = oracle.stokes3d_vep_solve(ref, oracle.rheology_struct(phases), _params(oracle, s, T_ghosted=True, cutoff=(1e-2, 1e2), viscosity_relaxation=0.1))
    stokes, pr, ρg = _upload(jr, s)
    T = from_numpy(s.arrays["T"], stokes.P.device)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, dict(T=T, P=stokes.P), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 25
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-8)
    assert np.ptp(ref["eta"]) > 1e-3
    out = _download(jr, stokes)
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-8, k
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    jr.compute_shear_heating_(thermal, stokes, pr, [dict(ph, shear_heat=1.0) for ph in phases], s.dt)
    sh = jr.to_numpy(thermal.shear_heating)
    assert (sh >= 0).all() and sh.max() > 0


def test_single_material_driver_with_power_law_creep_matches_oracle(jr, oracle):
    """Stokes2D.jl:345-557 with a DislocationCreep MaterialParams: compute_viscosity! then compute_viscosity_τII! every iteration, both on @strain(stokes)
    (Viscosity.jl:136-167)"""
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import VEP_MAP, _get
    from test_gpu_vep_extras import _nl_params
    s = jr.miniapps.thermal_convection2d(32, ar=1, iterMax=119, nout=40)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    ph = dict(s.extra["rheology"])
    # The in-loop compute_viscosity_τII! of this driver is handed the STRAIN-RATE invariant (the reference's _compute_viscosity! always passes @strain(stokes)),
    # which the law then reads as a stress; mirrored as written.  A is chosen so that this evaluation lands inside the cutoff at the strain rates the
    # convection reaches (~2e-17 / s), otherwise every cell sits on the upper cutoff and the comparison would say nothing about the law.
    n, E, R, Tm = 3.0, 150e3, 8.3145, 2285.0
    A = 0.5 * (2e-17) ** (1 - n) * np.exp(E / (R * Tm)) / ph.pop("eta")
    ph["creep"] = dict(kind="dislocation", A=A, n=n, E=E, V=0.0, R=R, apparatus="Invariant")
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_nonlinear_solve(ref, oracle.rheology_struct([ph]), _nl_params(oracle, s))
    dev = torch.device("cuda", torch.cuda.current_device())
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, path in VEP_MAP.items():
        _get(st, path).copy_(from_numpy(s.arrays[k], dev))
    ρg = (from_numpy(s.arrays["fx"], dev), from_numpy(s.arrays["fy"], dev))
    T = from_numpy(s.arrays["T"], dev)
    r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, ph, dict(T=T, P=st.P), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] == 120
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-7)
    assert np.ptp(np.log10(ref["eta"])) > 1.5 and (ref["eta"] > 1.001e16).sum() > ref["eta"].size // 4
    out = {k: jr.to_numpy(_get(st, path)) for k, path in VEP_MAP.items()}
    for k in out:
        assert max_rel_diff(out[k], ref[k]) <= 1e-7, k


def test_shearheating3d_reference_test_on_the_device(jr, oracle):
    """test/test_shearheating3D.jl:237-252 (particle-free restatement, jr.miniapps.shearheating3d): solve! with dt = Inf on the Duretz et al. dislocation-creep
    phases -> tensor_invariant! -> compute_dt -> compute_shear_heating!; the reference asserts `iters.err_evo1[end] < 1e-4` and a non-negative shear heating.
    The converged state is also compared with the oracle's (1e-6: two thousand iterations of a solve whose viscosity follows its own stress)."""
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep3d import _download, _params, _upload
    s = jr.miniapps.shearheating3d(16)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes3d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]), _params(oracle, s, T_ghosted=True, cutoff=s.kwargs["viscosity_cutoff"]))
    stokes, pr, ρg = _upload(jr, s)
    T = from_numpy(s.arrays["T"], stokes.P.device)
    phases = s.extra["phases"]
    jr.compute_viscosity_(stokes, pr, dict(T=T, P=stokes.P), phases, (-np.inf, np.inf))          # :127 (the solve repeats it with its cutoff)
    assert np.isfinite(jr.to_numpy(stokes.viscosity.η)).all()
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, dict(T=T, P=stokes.P), s.dt, None, kwargs=s.kwargs)
    assert r.err_evo1[-1] < 1e-4 and r.iter == r_ref["iter"]
    out = _download(jr, stokes)
    for k in ("Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "eta", "P"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-6, k
    jr.tensor_invariant_(stokes.ε)
    dt = jr.compute_dt_(stokes, s.extra["di"], s.extra["dt_diff"]) * 0.1
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    jr.compute_shear_heating_(thermal, stokes, pr, phases, dt)
    sh = jr.to_numpy(thermal.shear_heating)
    assert (sh >= 0).all() and sh.max() > 0                                                        # :251


def test_shearheating2d_reference_script_on_the_device(jr, oracle):
    """test/test_shearheating2D.jl:66-190 (particle-free, jr.miniapps.shearheating2d): 2D multiphase solve! with dt = Inf and dislocation-creep phases ->
    tensor_invariant! -> compute_dt -> compute_shear_heating! >= 0 (:246); state compared with the oracle's run of the same script."""
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.checks import max_rel_diff
    from test_gpu_vep2d import _download, _upload, _vep_params
    s = jr.miniapps.shearheating2d(32)
    ref = _cp(s.arrays)
    r_ref = oracle.stokes2d_vep_solve(ref, oracle.rheology_struct(s.extra["phases"]), _vep_params(oracle, s, T_ghosted=True, cutoff=s.kwargs["viscosity_cutoff"]))
    stokes, pr, ρg = _upload(jr, s)
    stokes.viscosity.ηv.fill_(1.0e20)
    T = from_numpy(s.arrays["T"], stokes.P.device)
    phases = s.extra["phases"]
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, phases, dict(T=T, P=stokes.P), s.dt, None, kwargs=s.kwargs)
    assert r.iter == r_ref["iter"] and np.isfinite(r.err_evo1).all()
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-5)
    out = _download(jr, stokes)
    for k in ("Vx", "Vy", "txx", "tyy", "txy", "eta", "P"):
        assert max_rel_diff(out[k], ref[k]) <= 1e-6, k
    jr.tensor_invariant_(stokes.ε)
    dt = jr.compute_dt_(stokes, s.extra["di"], s.extra["dt_diff"]) * 0.1
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    jr.compute_shear_heating_(thermal, stokes, pr, phases, dt)
    sh = jr.to_numpy(thermal.shear_heating)
    assert (sh >= 0).all() and sh.max() > 0
