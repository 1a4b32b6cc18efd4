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
