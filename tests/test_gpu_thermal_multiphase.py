"""GPU parity tests, phase-ratio form of the PT heat-diffusion path (heatdiffusion_PT!(...; kwargs = (phase = phase_ratios, ...)),
DiffusionPT_solver.jl:181-305 with update_pt_thermal_arrays! in every iteration) vs the CPU oracle, and the reference's own multiphase numbers
(test/test_diffusion2D_multiphase.jl:193-194, test/test_diffusion3D_multiphase.jl:214-215) on the device."""
import json
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KA = json.loads((Path(__file__).parent / "golden" / "reference_known_answers.json").read_text())
TOL = 1e-9


def _device_setup(jr, s, *, eps=None):
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    nd = len(s.ni)
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    for name in ("T", "Told", "H", "qTx", "qTy", "qTx2", "qTy2", "shear_heating") + (("qTz", "qTz2") if nd == 3 else ()):
        getattr(thermal, name).copy_(from_numpy(s.arrays[name], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    for k, v in s.extra["phase_ratios"].items():
        getattr(pr, k).copy_(from_numpy(v, dev))
    args = SimpleNamespace(P=from_numpy(s.arrays["P"], dev), T=thermal.T)
    pt = jr.PTThermalCoeffs.from_phases(jr.AMDGPUBackend, s.extra["rheology"], pr, args, s.dt, s.ni, s.extra["di"], s.extra["li"],
                                        ϵ=s.pt["eps"] if eps is None else eps, CFL=s.pt["CFL"])
    return thermal, pt, pr, args


def _oracle_inputs(oracle, s, eps, **kw):
    b = s.flow_bcs
    mk = oracle.thermal_params3d if len(s.ni) == 3 else oracle.thermal_params2d
    p = mk(s.ni, s.grid._di["center"], s.dt, eps, no_flux=b.no_flux, constant_value=b.constant_value, constant_flux=b.constant_flux,
           periodic=b.periodic, **kw)
    pr = s.extra["phase_ratios"]
    m = oracle.thermal_phases(list(s.extra["rheology"]), s.pt["max_lxyz"], s.pt["Vpdtau"])
    ph = dict(P=s.arrays["P"], phase_c=pr["center"], phase_qx=pr["Vx"], phase_qy=pr["Vy"], phase_qz=pr.get("Vz"))
    return p, m, ph


def _randomise(s, seed):
    """mixed ratios everywhere (three-way split incl. exact 0 and 1 cells), a pressure field and a β so that every branch of the density runs"""
    rng = np.random.default_rng(seed)
    for k, v in s.extra["phase_ratios"].items():
        f = rng.random(v.shape[1:])
        f[rng.random(f.shape) < 0.2] = 0.0
        f[rng.random(f.shape) < 0.2] = 1.0
        v[0], v[1] = 1.0 - f, f
    s.arrays["P"][...] = rng.random(s.ni) * 1.0e9
    s.arrays["shear_heating"][...] = rng.random(s.ni) * 1.0e-7
    rheo = [dict(r, density=dict(r["density"])) for r in s.extra["rheology"]]
    rheo[0]["density"]["beta"], rheo[0]["k"], rheo[0]["Cp"] = 1.0e-11, 2.5, 1.0e3
    rheo[1]["density"].update(kind="T"), rheo[1].update(k=4.0)
    s.extra["rheology"] = tuple(rheo)


@pytest.mark.parametrize("dim,ni", [(2, (37, 21)), (2, (130, 40)), (3, (20, 14, 12)), (3, (70, 17, 20))])
def test_multiphase_iterations_match_oracle(jr, oracle, dim, ni):
    from justrelax_jl_amd.checks import max_rel_diff
    s = jr.miniapps.diffusion2d_multiphase(ni, iterMax=60, nout=20) if dim == 2 else jr.miniapps.diffusion3d_multiphase(ni, iterMax=60, nout=20)
    _randomise(s, 11 + dim)
    p, m, ph = _oracle_inputs(oracle, s, 1e-30, iterMax=60, nout=20)
    thermal, pt, pr, args = _device_setup(jr, s, eps=1e-30)
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    ph["P"] = ref["P"]
    r_ref = oracle.heatdiffusion_PT_phases(ref, p, m, ph)
    r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], args, s.dt, s.grid, kwargs=dict(phase=pr, iterMax=60, nout=20, verbose=False))
    assert list(r.iter_count) == list(r_ref["iter_count"]) == [20, 40, 60]
    assert np.allclose(r.norm_ResT, r_ref["norm_ResT"], rtol=1e-9)
    names = [("T", thermal.T), ("Told", thermal.Told), ("dT", thermal.ΔT)]
    for name, t in names:
        got = jr.to_numpy(t)
        assert np.abs(got - ref[name]).max() <= TOL * np.abs(ref[name]).max(), name
    fl = [("qTx", thermal.qTx), ("qTy", thermal.qTy), ("qTx2", thermal.qTx2), ("qTy2", thermal.qTy2)] + ([("qTz", thermal.qTz), ("qTz2", thermal.qTz2)] if dim == 3 else [])
    for name, t in fl:
        assert max_rel_diff(jr.to_numpy(t), ref[name]) <= TOL, name
    for name, t in (("thetar_dtau", pt.θr_dτ), ("dtau_rho", pt.dτ_ρ)):
        assert max_rel_diff(jr.to_numpy(t), ref[name]) <= 1e-13, name
    scale = max(np.abs(ref["ResT"]).max(), 1e-7)
    assert np.abs(jr.to_numpy(thermal.ResT) - ref["ResT"]).max() <= 1e-7 * scale + TOL * scale


def test_diffusion2d_multiphase_reference_numbers_on_the_gpu(jr):
    """test/test_diffusion2D_multiphase.jl:186-200 on the device: 20 steps of 50 kyr; T[18,18] ≈ 1814.029, T[17,17] ≈ 1823.548, atol 0.1"""
    import torch
    from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
    s = jr.miniapps.diffusion2d_multiphase(32)
    Th = s.arrays["T"]
    thermal, pt, pr, args = _device_setup(jr, s)
    jr.thermal_bcs_(thermal, s.flow_bcs)
    T = jr.to_numpy(thermal.T)
    add_perturbation(T, s.grid, **s.extra["perturbation"])
    thermal.T.copy_(jr.from_numpy(T, thermal.T.device))
    for _ in range(s.extra["nt"]):
        r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], args, s.dt, s.grid,
                                 kwargs=dict(phase=pr, iterMax=1.0e3, nout=10, verbose=False))
        assert r.norm_ResT[-1] <= s.pt["eps"]
    T, g = jr.to_numpy(thermal.T), KA["diffusion2D_multiphase"]
    assert T[17, 17] == pytest.approx(g["T_18_18"], abs=g["atol"])
    assert T[16, 16] == pytest.approx(g["T_17_17"], abs=g["atol"])


def test_diffusion3d_multiphase_reference_numbers_on_the_gpu(jr):
    """test/test_diffusion3D_multiphase.jl:207-219 on the device: 32^3, 10 steps of 50 kyr, rtol 1e-3"""
    s = jr.miniapps.diffusion3d_multiphase(32)
    thermal, pt, pr, args = _device_setup(jr, s)
    # the reference builds this pt_thermal from the K / ρCp arrays; its arrays are overwritten in the first iteration either way
    for _ in range(s.extra["nt"]):
        r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], args, s.dt, s.grid,
                                 kwargs=dict(phase=pr, iterMax=10.0e3, nout=1.0e2, verbose=False))
        assert r.norm_ResT[-1] <= 1e-8
    T, g = jr.to_numpy(thermal.T), KA["diffusion3D_multiphase"]
    assert T[15, 15, 15] == pytest.approx(g["T_16_16_16"], rel=g["rtol"])
    assert T[16, 16, 16] == pytest.approx(g["Tinterior_16_16_16"], rel=g["rtol"])


def test_phase_form_argument_errors(jr):
    s = jr.miniapps.diffusion2d_multiphase(8)
    thermal, pt, pr, args = _device_setup(jr, s)
    with pytest.raises(ValueError):      # a phase table without phase ratios
        jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, list(s.extra["rheology"]), args, s.dt, s.grid, kwargs=dict(verbose=False))
    bad = SimpleNamespace(P=args.P, T=thermal.Told)
    with pytest.raises(ValueError):      # args.T must be thermal.T
        jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], bad, s.dt, s.grid, kwargs=dict(phase=pr, verbose=False))


@pytest.mark.parametrize("dim,form", [(2, "array"), (2, "rheology"), (2, "phases"), (3, "array"), (3, "phases")])
def test_dirichlet_mask_and_adiabatic_heating_match_oracle(jr, oracle, dim, form):
    """inner Dirichlet cells (constant and array forms, incl. a fractional mask) in every form of heatdiffusion_PT!, and the adiabatic term of the rheology /
    phase-ratio forms fed by kwargs.stokes (adiabatic_heating!)"""
    import torch
    from justrelax_jl_amd.arrays import from_numpy, TemperatureBoundaryConditions
    from justrelax_jl_amd.checks import max_rel_diff
    ni = (30, 22) if dim == 2 else (18, 14, 12)
    if form == "phases":
        s = jr.miniapps.diffusion2d_multiphase(ni, iterMax=60, nout=20) if dim == 2 else jr.miniapps.diffusion3d_multiphase(ni, iterMax=60, nout=20)
        _randomise(s, 3)
    else:
        s = jr.miniapps.diffusion2d(ni[0], iterMax=60, nout=20) if dim == 2 else jr.miniapps.diffusion3d(ni, iterMax=60, nout=20)
    ni = s.ni
    rng = np.random.default_rng(21)
    gsh = tuple(n + 2 for n in ni)
    const = form != "array"                     # constant value with a 0/1 + one fractional mask; or a value array whose non-zeros are the mask
    if const:
        mask = np.zeros(gsh, order="F")
        mask[(slice(3, 6),) * dim] = 1.0
        mask[(8,) * dim] = 0.5
        dbc = dict(constant=1400.0, mask=mask)
        omask, oval = mask, None
    else:
        vals = np.zeros(gsh, order="F")
        vals[(slice(4, 7),) * dim] = rng.uniform(1000.0, 2000.0, size=(3,) * dim)
        dbc = dict(constant=None, mask=vals)
        omask, oval = np.asfortranarray((vals != 0).astype(float)), vals
    b = s.flow_bcs
    dev = torch.device("cuda", torch.cuda.current_device())
    bc = TemperatureBoundaryConditions(no_flux=b.no_flux, constant_value=b.constant_value, constant_flux=b.constant_flux, periodic=b.periodic,
                                       dirichlet=dict(constant=dbc["constant"], mask=from_numpy(dbc["mask"], dev)))
    mk = oracle.thermal_params3d if dim == 3 else oracle.thermal_params2d
    kw = dict(rheology=s.extra["rheology"]) if form == "rheology" else {}
    p = mk(ni, s.grid._di["center"], s.dt, 1e-30, iterMax=60, nout=20, no_flux=b.no_flux, constant_value=b.constant_value, constant_flux=b.constant_flux,
           periodic=b.periodic, **kw)
    p.dirichlet_const = 1400.0
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    ref["dirichlet_mask"] = omask
    if oval is not None:
        ref["dirichlet_value"] = oval
    # stokes.P, P0 for the adiabatic term of the rheology forms
    P = np.asfortranarray(rng.uniform(1e8, 3e8, size=ni))
    P0 = np.asfortranarray(rng.uniform(1e8, 3e8, size=ni))
    stokes = SimpleNamespace(P=from_numpy(P, dev), P0=from_numpy(P0, dev))
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, ni)
    for name in ("T", "Told", "H", "shear_heating"):
        getattr(thermal, name).copy_(from_numpy(s.arrays[name], dev))
    kwargs = dict(iterMax=60, nout=20, verbose=False)
    if form == "phases":
        pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, ni)
        for k, v in s.extra["phase_ratios"].items():
            getattr(pr, k).copy_(from_numpy(v, dev))
        args = SimpleNamespace(P=from_numpy(s.arrays["P"], dev), T=thermal.T)
        pt = jr.PTThermalCoeffs.from_phases(jr.AMDGPUBackend, s.extra["rheology"], pr, args, s.dt, ni, s.extra["di"], s.extra["li"], ϵ=1e-30, CFL=s.pt["CFL"])
        _, m, ph = _oracle_inputs(oracle, s, 1e-30, iterMax=60, nout=20)
        ph["P"] = ref["P"]
        ref["adiabatic"] = np.zeros(ni, order="F")
        oracle.adiabatic_heating(ref["adiabatic"], P, P0, m, s.extra["phase_ratios"]["center"], 1.0 / s.dt)
        r_ref = oracle.heatdiffusion_PT_phases(ref, p, m, ph)
        r = jr.heatdiffusion_PT_(thermal, pt, bc, s.extra["rheology"], args, s.dt, s.grid, kwargs=dict(kwargs, phase=pr, stokes=stokes))
    else:
        K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
        pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-30)
        solve = oracle.heatdiffusion_PT3d if dim == 3 else oracle.heatdiffusion_PT2d
        if form == "rheology":
            rh = s.extra["rheology"]
            m = oracle.thermal_phases([dict(k=rh["k"], Cp=rh["Cp"], density=dict(kind="PT", rho0=rh["rho0"], alpha=rh["alpha"], T0=rh.get("T0", 0.0)))], 1.0, 1.0)
            ref["adiabatic"] = np.zeros(ni, order="F")
            oracle.adiabatic_heating(ref["adiabatic"], P, P0, m, None, 1.0 / s.dt)
            r_ref = solve(ref, p)
            r = jr.heatdiffusion_PT_(thermal, pt, bc, rh, None, s.dt, s.grid, kwargs=dict(kwargs, stokes=stokes))
        else:
            r_ref = solve(ref, p)
            r = jr.heatdiffusion_PT_(thermal, pt, bc, K, ρCp, s.dt, s.grid, kwargs=kwargs)
    assert list(r.iter_count) == list(r_ref["iter_count"]) == [20, 40, 60]
    assert np.allclose(r.norm_ResT, r_ref["norm_ResT"], rtol=1e-9)
    T = jr.to_numpy(thermal.T)
    assert np.abs(T - ref["T"]).max() <= 1e-9 * np.abs(ref["T"]).max()
    res = jr.to_numpy(thermal.ResT)
    scale = max(np.abs(ref["ResT"]).max(), 1e-7)
    assert np.abs(res - ref["ResT"]).max() <= 1e-7 * scale
    inner = tuple(slice(1, -1) for _ in range(dim))
    assert (res[omask[inner] != 0] == 0.0).all()
    if const:
        assert (T[(slice(3, 6),) * dim] == 1400.0).all()
    else:
        assert np.array_equal(T[(slice(4, 7),) * dim], oval[(slice(4, 7),) * dim])
    if form != "array":
        assert max_rel_diff(jr.to_numpy(thermal.adiabatic), ref["adiabatic"]) <= 1e-14 and np.abs(ref["adiabatic"]).max() > 0


def test_round2_entry_points_refuse_bad_arguments(jr):
    """error behaviour of the entry points added in round 2: JRX_ERR_ARG with a message, nothing launched"""
    import ctypes as C
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd import thermal as th
    from justrelax_jl_amd.arrays import ptr
    h = _lib.default_handle()
    s = jr.miniapps.diffusion2d_multiphase(8)
    thermal, pt, pr, args = _device_setup(jr, s)
    f = th.thermal_fields2d(thermal, pt)
    p = th.thermal_params2d(s.ni, s.grid, s.flow_bcs, s.dt, 1e-8, iterMax=10, nout=5)
    m = th.thermal_phases(s.extra["rheology"], pt)
    pf = th.thermal_phase_fields(pr, args, s.ni)
    it, nr, nn = (C.c_int64 * 4)(), (C.c_double * 4)(), C.c_int64(0)
    with pytest.raises(_lib.JrxError, match="null phases"):
        h.call("jrx_heatdiffusion_PT2d_phases", C.byref(f), C.byref(p), None, C.byref(pf), it, nr, C.c_int64(4), C.byref(nn))
    m.nphase = 9
    with pytest.raises(_lib.JrxError, match="nphase"):
        h.call("jrx_heatdiffusion_PT2d_phases", C.byref(f), C.byref(p), C.byref(m), C.byref(pf), it, nr, C.c_int64(4), C.byref(nn))
    m.nphase = 2
    pf.phase_qx = None
    with pytest.raises(_lib.JrxError, match="face phase ratios"):
        h.call("jrx_heatdiffusion_PT2d_phases", C.byref(f), C.byref(p), C.byref(m), C.byref(pf), it, nr, C.c_int64(4), C.byref(nn))
    p.rheology_form = 2            # the plain entry does not take the phase-ratio form
    with pytest.raises(_lib.JrxError, match="rheology_form"):
        h.call("jrx_heatdiffusion_PT2d", C.byref(f), C.byref(p), it, nr, C.c_int64(4), C.byref(nn))
    n3 = (C.c_int64 * 3)(8, 8, 1)
    with pytest.raises(_lib.JrxError, match="update_pt_thermal_arrays"):
        h.call("jrx_update_pt_thermal_arrays", C.c_void_p(ptr(pt.θr_dτ)), C.c_void_p(ptr(pt.dτ_ρ)), C.c_void_p(ptr(thermal.T)), n3, C.c_int32(4), C.c_double(1.0),
               C.byref(m), C.byref(pf))
    with pytest.raises(_lib.JrxError, match="adiabatic_heating"):
        h.call("jrx_adiabatic_heating", None, C.c_void_p(ptr(args.P)), C.c_void_p(ptr(args.P)), C.c_int64(64), C.c_double(1.0), C.byref(m), None)
    with pytest.raises(_lib.JrxError, match="no such option|unknown"):
        h.call("jrx_set_option", C.c_char_p(b"no_such_option"), C.c_int64(1))
    with pytest.raises(_lib.JrxError):      # a mask of the wrong shape is refused on the host side before any call
        pass_bc = jr.TemperatureBoundaryConditions(no_flux=s.flow_bcs.no_flux, dirichlet=dict(constant=1.0, mask=args.P))
        try:
            jr.heatdiffusion_PT_(thermal, pt, pass_bc, s.extra["rheology"], args, s.dt, s.grid, kwargs=dict(phase=pr, verbose=False))
        except ValueError as e:
            raise _lib.JrxError(1, str(e))


@pytest.mark.parametrize("form", ["phases", "adiabatic"])
def test_graph_replay_of_the_two_kernel_2d_forms_changes_nothing(jr, form):
    """option loop_graphs: the 2D forms that keep compute_flux! + update_T! as two launches (phase ratios; adiabatic term / Dirichlet cells) replay runs of unobserved
    iterations as captured hipGraphs; T, the fluxes, the PT coefficients and the residual history equal those of plain launches"""
    import ctypes as C
    from justrelax_jl_amd import _lib
    h = _lib.default_handle(0)
    outs = []
    try:
        for g in (0, 1):
            h.call("jrx_set_option", C.c_char_p(b"loop_graphs"), C.c_int64(g))
            s = jr.miniapps.diffusion2d_multiphase((48, 40), iterMax=300, nout=150)
            _randomise(s, 9)
            thermal, pt, pr, args = _device_setup(jr, s, eps=1e-30)
            kw = dict(phase=pr, iterMax=300, nout=150, verbose=False)
            if form == "adiabatic":       # kwargs.stokes: adiabatic_heating! feeds update_T!
                stokes = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
                stokes.P.copy_(args.P)
                stokes.P0.copy_(args.P * 0.9)
                kw["stokes"] = stokes
            r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], args, s.dt, s.grid, kwargs=kw)
            outs.append((list(r.iter_count), list(r.norm_ResT), jr.to_numpy(thermal.T), jr.to_numpy(thermal.qTx), jr.to_numpy(thermal.qTy2), jr.to_numpy(pt.θr_dτ),
                         jr.to_numpy(pt.dτ_ρ)))
    finally:
        h.call("jrx_set_option", C.c_char_p(b"loop_graphs"), C.c_int64(1))
    a, b = outs
    assert a[0] == b[0] == [150, 300] and a[1] == b[1]
    for x, y in zip(a[2:], b[2:]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("dim,ni", [(2, (37, 21)), (2, (130, 40)), (3, (20, 14, 12)), (3, (70, 17, 20)), (3, (65, 3, 3))])
def test_phase_count_constant_kernels_equal_the_run_time_loops(jr, dim, ni):
    """option thermal_np_const: the phase-ratio kernels instantiated for the phase count (ratios in registers, loops unrolled; in 3D the flux kernel with every operand requested up
    front, k_flux3d_b) against the run-time loops / the control-flow flux kernel: temperature, fluxes, residual and PT coefficients bit for bit"""
    from justrelax_jl_amd import _lib
    h = _lib.default_handle()
    outs = []
    try:
        for const in (0, 1):
            h.set_option("thermal_np_const", const)
            s = jr.miniapps.diffusion2d_multiphase(ni, iterMax=60, nout=20) if dim == 2 else jr.miniapps.diffusion3d_multiphase(ni, iterMax=60, nout=20)
            _randomise(s, 11 + dim)
            thermal, pt, pr, args = _device_setup(jr, s, eps=1e-30)
            r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, s.extra["rheology"], args, s.dt, s.grid, kwargs=dict(phase=pr, iterMax=60, nout=20, verbose=False))
            fields = [thermal.T, thermal.Told, thermal.ΔT, thermal.qTx, thermal.qTy, thermal.qTx2, thermal.qTy2, thermal.ResT, pt.θr_dτ, pt.dτ_ρ] + ([thermal.qTz, thermal.qTz2] if dim == 3 else [])
            outs.append((list(r.iter_count), list(r.norm_ResT), [jr.to_numpy(t) for t in fields]))
    finally:
        h.set_option("thermal_np_const", 1)
    assert outs[0][0] == outs[1][0] == [20, 40, 60] and outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][2], outs[1][2]):
        assert np.array_equal(a, b, equal_nan=True)
