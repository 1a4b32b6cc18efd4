"""Operator API of the PT heat-diffusion path: heatdiffusion_PT_, thermal_bcs_.

Reference: src/thermal_diffusion/DiffusionPT_solver.jl:11-17 (dispatch), :34-149 (array form),
:181-305 (rheology form); src/boundaryconditions/BoundaryConditions.jl:39-53.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .arrays import ptr
from .grid import Geometry, legacy_uniform_grid
from .stokes import _require_gpu

_ORDER = ("left", "right", "top", "bot")


_ORDER3 = ("left", "right", "front", "back", "top", "bot")


def thermal_params2d(ni, grid, thermal_bc, dt, ϵ, *, iterMax=50_000, nout=1000, verbose=True, rheology=None, **_):
    """jrx_thermal2d_params / jrx_thermal3d_params (by len(ni)) from the reference's keyword arguments"""
    nonuni = getattr(grid, "nonuniform", False)
    if nonuni and len(ni) == 3:
        raise NotImplementedError("non-uniform grids are built for the 2D drivers only")
    _di = tuple(float(n) / float(l) for n, l in zip(grid.ni, grid.li)) if nonuni else grid._di["center"]      # scalars unused on a non-uniform grid
    if len(ni) == 3:
        p = _lib.Thermal3DParams()
        p.nx, p.ny, p.nz, p._dx, p._dy, p._dz, p.dt, p.eps = ni[0], ni[1], ni[2], _di[0], _di[1], _di[2], float(dt), float(ϵ)
        order = _ORDER3
    else:
        p = _lib.Thermal2DParams()
        p.nx, p.ny, p._dx, p._dy, p.dt, p.eps = ni[0], ni[1], _di[0], _di[1], float(dt), float(ϵ)
        order = _ORDER
    p.iterMax, p.nout, p.verbose = int(iterMax), int(nout), int(bool(verbose))
    for i, k in enumerate(order):
        p.no_flux[i] = int(bool(thermal_bc.no_flux.get(k, False)))
        p.periodic[i] = int(bool(thermal_bc.periodic.get(k, False)))
        v = thermal_bc.constant_value.get(k, False)
        p.constant_value_on[i] = int(v is not False and v is not None)         # `bc.bot === false ? ... : 2*bc.bot - T`
        p.constant_value[i] = float(v) if p.constant_value_on[i] else 0.0
        v = thermal_bc.constant_flux.get(k, False)
        on = not isinstance(v, bool) and v is not None                         # `!isa(bc_flux.left, Bool)`
        p.constant_flux_on[i] = int(on)
        p.constant_flux[i] = float(v) if on else 0.0
    dbc = getattr(thermal_bc, "dirichlet", None)
    if dbc is not None and dbc.get("constant") is not None:
        p.dirichlet_const = float(dbc["constant"])
    if rheology is not None:
        p.rheology_form = 1
        p.k_const, p.Cp, p.rho0, p.alpha, p.T0 = (rheology["k"], rheology["Cp"], rheology["rho0"], rheology["alpha"],
                                                   rheology.get("T0", 0.0))
    if nonuni:          # _di.center for compute_flux!, _di.vertex for update_T! / check_res! (DiffusionPT_solver.jl:243-282)
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        d = grid._di
        arrs = tuple(torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev) for a in (d["center"][0], d["center"][1], d["vertex"][0], d["vertex"][1]))
        p._spacing_keepalive = arrs
        for q, a in enumerate(arrs):
            p.inv_spacing[q] = a.data_ptr()
    return p


def _dirichlet_arrays(thermal, thermal_bc):
    """(mask, value) of thermal_bc.dirichlet = dict(constant=number | None, mask=array (ni .+ 2)) -- Dirichlet(constant, mask), Dirichlet.jl:131-135:
    with constant = None the array holds the values and its non-zero entries are the mask (DirichletBoundaryCondition(A), :15-20)"""
    dbc = getattr(thermal_bc, "dirichlet", None) if thermal_bc is not None else None
    if not dbc or dbc.get("mask") is None:
        return None, None
    m = dbc["mask"]
    if tuple(m.shape) != tuple(thermal.T.shape):
        raise ValueError(f"the Dirichlet mask must have the shape of thermal.T {tuple(thermal.T.shape)}")
    if dbc.get("constant") is not None:
        return m, None
    t = torch.empty_like(m)                        # Mask(copy(A)) filled with T.(.!iszero.(A)); keeps the column-major strides
    t.copy_((m != 0).to(m.dtype))
    return t, m


def thermal_fields2d(thermal, pt_thermal, K=None, ρCp=None, thermal_bc=None, adiabatic=None):
    """jrx_thermal2d_fields / jrx_thermal3d_fields (by the dimension of `thermal`)"""
    three = len(thermal._ni) == 3
    f = _lib.Thermal3DFields() if three else _lib.Thermal2DFields()
    vals = dict(T=thermal.T, Told=thermal.Told, dT=thermal.ΔT, qTx=thermal.qTx, qTx2=thermal.qTx2, qTy=thermal.qTy,
                qTy2=thermal.qTy2, H=thermal.H, shear_heating=thermal.shear_heating, ResT=thermal.ResT, K=K, rhoCp=ρCp,
                thetar_dtau=pt_thermal.θr_dτ, dtau_rho=pt_thermal.dτ_ρ)
    if three:
        vals.update(qTz=thermal.qTz, qTz2=thermal.qTz2)
    dm, dv = _dirichlet_arrays(thermal, thermal_bc)
    vals.update(adiabatic=adiabatic, dirichlet_mask=dm, dirichlet_value=dv)
    for n in (_lib.T3_NAMES if three else _lib.T2_NAMES) + _lib.T_OPT:
        setattr(f, n, ptr(vals.get(n)))
    f._keep = vals
    return f


_RHO_KINDS = {"constant": 0, "PT": 1, "T": 2, "compressible": 3}


def thermal_phases(rheology, pt_thermal):
    """jrx_thermal_phases from the per-phase material table: a sequence of dict(k, Cp, Hr=0, density=dict(kind, rho0, alpha, beta, T0, P0))
    (ConstantConductivity, ConstantHeatCapacity, ConstantRadioactiveHeat and a density law per phase) plus pt_thermal.max_lxyz / Vpdτ"""
    if len(rheology) > _lib.MAXPHASE:
        raise ValueError(f"at most {_lib.MAXPHASE} phases")
    m = _lib.ThermalPhases()
    m.nphase = len(rheology)
    for q, ph in enumerate(rheology):
        m.k[q], m.Cp[q], m.Hr[q] = float(ph["k"]), float(ph["Cp"]), float(ph.get("Hr", 0.0))
        d = ph.get("density", {})
        m.rho_kind[q] = _RHO_KINDS[d.get("kind", "constant")]
        m.rho0[q], m.alpha[q], m.beta[q], m.T0[q], m.P0[q] = (float(d.get(k, 0.0)) for k in ("rho0", "alpha", "beta", "T0", "P0"))
    m.max_lxyz, m.Vpdtau = float(pt_thermal.max_lxyz), float(pt_thermal.Vpdτ)
    return m


def thermal_phase_fields(phase_ratios, args, ni):
    """jrx_thermal_phase_fields: args.P and phase_ratios.center / .Vx / .Vy [/ .Vz]"""
    P = args["P"] if isinstance(args, dict) else args.P
    if tuple(P.shape) != tuple(ni):
        raise ValueError(f"args.P must have the shape of the cell centres {tuple(ni)}")
    f = _lib.ThermalPhaseFields()
    vals = dict(P=P, phase_c=phase_ratios.center, phase_qx=phase_ratios.Vx, phase_qy=phase_ratios.Vy,
                phase_qz=getattr(phase_ratios, "Vz", None) if len(ni) == 3 else None)
    for n in _lib.TPH_NAMES:
        setattr(f, n, ptr(vals[n]))
    f._keep = vals
    return f


def update_pt_thermal_arrays_(pt_thermal, phase_ratios, rheology, args, _dt, *, handle=None):
    """update_pt_thermal_arrays!(pt_thermal, phase_ratios, rheology, args, _dt) -- DiffusionPT_coefficients.jl:105-121"""
    T = args["T"] if isinstance(args, dict) else args.T
    _require_gpu(T)
    ni = tuple(pt_thermal.θr_dτ.shape)
    if tuple(T.shape) != tuple(n + 2 for n in ni):
        raise ValueError("args.T must be thermal.T (ni .+ 2)")
    h = handle or _lib.default_handle(T.device.index)
    m, f = thermal_phases(rheology, pt_thermal), thermal_phase_fields(phase_ratios, args, ni)
    n3 = (C.c_int64 * 3)(*(tuple(ni) + (1,) * (3 - len(ni))))
    torch.cuda.current_stream(T.device).synchronize()
    h.call("jrx_update_pt_thermal_arrays", C.c_void_p(ptr(pt_thermal.θr_dτ)), C.c_void_p(ptr(pt_thermal.dτ_ρ)), C.c_void_p(ptr(T)), n3,
           C.c_int32(len(ni)), C.c_double(1.0 / _dt), C.byref(m), C.byref(f))


def heatdiffusion_PT_(thermal, pt_thermal, thermal_bc, A, B, dt, grid_or_di, *, kwargs=None, handle=None):
    """heatdiffusion_PT!(thermal, pt_thermal, thermal_bc, K, ρCp, dt, grid; kwargs)          [array form]
       heatdiffusion_PT!(thermal, pt_thermal, thermal_bc, rheology, args, dt, grid; kwargs)  [rheology form]

    In the rheology form `A` is a dict(k, Cp, rho0, alpha, T0) (constant conductivity / heat capacity and a
    PT_Density) and `B` the reference's `args` (ignored: T is thermal.T, P does not enter with β = 0).
    With kwargs["phase"] = PhaseRatios, `A` is the per-phase table (see thermal_phases) and `B` = args with P (ni) and T = thermal.T.
    Returns (iter_count, norm_ResT) like DiffusionPT_solver.jl:148."""
    _require_gpu(thermal)
    kw = dict(kwargs or {})
    ni = thermal._ni
    grid = grid_or_di if isinstance(grid_or_di, Geometry) else legacy_uniform_grid(ni, grid_or_di)
    h = handle or _lib.default_handle(thermal.T.device.index)
    phase = kw.pop("phase", None)
    stokes = kw.pop("stokes", None)
    adiabatic = None
    if stokes is not None and isinstance(A, (dict, list, tuple)):
        # adiabatic_heating!(thermal, stokes, rheology, phases, _dt) (DiffusionPT_solver.jl:211-213): thermal.adiabatic = (P - P0) α / dt
        table = [dict(k=A["k"], Cp=A["Cp"], density=dict(kind="PT", rho0=A["rho0"], alpha=A["alpha"], T0=A.get("T0", 0.0)))] if isinstance(A, dict) else list(A)
        m_ad = thermal_phases(table, SimpleNamespace(max_lxyz=1.0, Vpdτ=1.0))
        torch.cuda.current_stream(thermal.T.device).synchronize()
        h.call("jrx_adiabatic_heating", C.c_void_p(ptr(thermal.adiabatic)), C.c_void_p(ptr(stokes.P)), C.c_void_p(ptr(stokes.P0)),
               C.c_int64(int(np.prod(ni))), C.c_double(float(dt)), C.byref(m_ad), C.c_void_p(ptr(phase.center) if phase is not None else 0))
        adiabatic = thermal.adiabatic
    if phase is not None:
        if isinstance(A, dict):
            A = [A]
        T = B["T"] if isinstance(B, dict) else B.T
        if T.data_ptr() != thermal.T.data_ptr():
            raise ValueError("args.T must be thermal.T in the phase-ratio form")
        m, pf = thermal_phases(A, pt_thermal), thermal_phase_fields(phase, B, ni)
        p = thermal_params2d(ni, grid, thermal_bc, dt, pt_thermal.ϵ, **kw)
        f = thermal_fields2d(thermal, pt_thermal, thermal_bc=thermal_bc, adiabatic=adiabatic)
        cap = int(p.iterMax // p.nout + 2)
        it, nr, nn = np.zeros(cap, dtype=np.int64), np.zeros(cap), C.c_int64(0)
        torch.cuda.current_stream(thermal.T.device).synchronize()
        h.call("jrx_heatdiffusion_PT3d_phases" if len(ni) == 3 else "jrx_heatdiffusion_PT2d_phases", C.byref(f), C.byref(p), C.byref(m), C.byref(pf),
               it.ctypes.data_as(C.POINTER(C.c_int64)), nr.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(cap), C.byref(nn))
        return SimpleNamespace(iter_count=it[: nn.value].copy(), norm_ResT=nr[: nn.value].copy())
    if isinstance(A, (list, tuple)):
        raise ValueError("a multi-phase rheology needs kwargs['phase'] = PhaseRatios")
    if isinstance(A, dict):
        p = thermal_params2d(ni, grid, thermal_bc, dt, pt_thermal.ϵ, rheology=A, **kw)
        f = thermal_fields2d(thermal, pt_thermal, thermal_bc=thermal_bc, adiabatic=adiabatic)
    else:
        p = thermal_params2d(ni, grid, thermal_bc, dt, pt_thermal.ϵ, **kw)
        f = thermal_fields2d(thermal, pt_thermal, A, B, thermal_bc=thermal_bc)
    cap = int(p.iterMax // p.nout + 2)
    it, nr, nn = np.zeros(cap, dtype=np.int64), np.zeros(cap), C.c_int64(0)
    torch.cuda.current_stream(thermal.T.device).synchronize()
    h.call("jrx_heatdiffusion_PT3d" if len(ni) == 3 else "jrx_heatdiffusion_PT2d", C.byref(f), C.byref(p), it.ctypes.data_as(C.POINTER(C.c_int64)),
           nr.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(cap), C.byref(nn))
    return SimpleNamespace(iter_count=it[: nn.value].copy(), norm_ResT=nr[: nn.value].copy())


def thermal_bcs_(thermal_or_T, thermal_bc, *, handle=None):
    """thermal_bcs!(thermal, bcs) -- BoundaryConditions.jl:39-53"""
    T = thermal_or_T.T if hasattr(thermal_or_T, "Told") else thermal_or_T
    _require_gpu(T)
    h = handle or _lib.default_handle(T.device.index)
    ni = tuple(n - 2 for n in T.shape)
    fake = SimpleNamespace(_di=dict(center=(1.0,) * T.dim()))
    p = thermal_params2d(ni, fake, thermal_bc, 1.0, 0.0)
    torch.cuda.current_stream(T.device).synchronize()
    h.call("jrx_thermal_bcs3d" if T.dim() == 3 else "jrx_thermal_bcs2d", C.c_void_p(ptr(T)), C.byref(p))


def thermal_iteration_(thermal, pt_thermal, thermal_bc, A, B, dt, grid, *, check_res=False, handle=None):
    """One PT iteration (compute_flux! + update_T! + thermal_bcs!), optionally followed by check_res!."""
    _require_gpu(thermal)
    h = handle or _lib.default_handle(thermal.T.device.index)
    ni = thermal._ni
    if isinstance(A, dict):
        p = thermal_params2d(ni, grid, thermal_bc, dt, pt_thermal.ϵ, rheology=A)
        f = thermal_fields2d(thermal, pt_thermal, thermal_bc=thermal_bc)
    else:
        p = thermal_params2d(ni, grid, thermal_bc, dt, pt_thermal.ϵ)
        f = thermal_fields2d(thermal, pt_thermal, A, B, thermal_bc=thermal_bc)
    torch.cuda.current_stream(thermal.T.device).synchronize()
    d = "3d" if len(ni) == 3 else "2d"
    h.call(f"jrx_thermal{d}_iteration", C.byref(f), C.byref(p))
    if check_res:
        h.call(f"jrx_thermal{d}_check_res", C.byref(f), C.byref(p))
