"""Operator API of the Stokes hot path -- the methods a backend adds to JustRelax's generics.

Reference method table being mirrored (src/ext/AMDGPU/3D.jl:205-239,397-399; 2D.jl): solve!,
flow_bcs!, velocity2displacement!, compute_maxloc!.  Julia's `f!` is spelled `f_`.
Every function forwards to one C-ABI entry point of include/jrx.h; nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .arrays import StokesArrays, fzeros, ptr
from .backend import AMDGPUBackendTrait, CPUBackendTrait, backend
from .grid import Geometry, global_grid, legacy_uniform_grid


def _require_gpu(x):
    tr = backend(x)
    if isinstance(tr, CPUBackendTrait):
        raise NotImplementedError(
            "CPU arrays: this package implements only the AMDGPU (HIP, gfx950) backend; the CPU backend is the "
            "reference's own JustRelax_CPU.  There is no CPU fallback.")
    if not isinstance(tr, AMDGPUBackendTrait):
        raise ValueError("Backend not supported")
    return tr


def _as_field(x, ni, dev):
    if isinstance(x, torch.Tensor):
        return x
    return fzeros(ni, dev, float(x))


def _center_inv(grid):
    _di = getattr(grid, "_di", None)
    if getattr(grid, "nonuniform", False):          # the scalars are unused then (every stencil reads its spacing array); keep them finite
        return tuple(float(n) / float(l) for n, l in zip(grid.ni, grid.li))
    return _di["center"] if isinstance(_di, dict) else _di


def _set_spacing2d(p, grid, stokes):
    """non-uniform Geometry: hand the six inverse-spacing arrays to the C ABI (kept alive on the params object)"""
    if not getattr(grid, "nonuniform", False):
        return
    arrs = grid.inv_spacing2d(stokes.P.device)
    p._spacing_keepalive = arrs
    for q, a in enumerate(arrs):
        p.inv_spacing[q] = a.data_ptr()


def _as_grid(stokes, grid_or_di):
    if isinstance(grid_or_di, Geometry):
        return grid_or_di
    return legacy_uniform_grid(stokes._ni, grid_or_di)      # Stokes3D.jl:188-203 / Stokes2D.jl:165-178


def _opt(obj, name):
    return obj.__dict__.get(name)


def fields3d(stokes: StokesArrays, ρg, K, G) -> _lib.Stokes3DFields:
    f = _lib.Stokes3DFields()
    s = stokes
    vals = dict(P=s.P, P0=s.P0, divV=s.divV, Q=s.Q, Vx=s.V.Vx, Vy=s.V.Vy, Vz=s.V.Vz, Ux=s.U.Ux, Uy=s.U.Uy, Uz=s.U.Uz,
                eta=s.viscosity.η, K=K, G=G, fx=ρg[0], fy=ρg[1], fz=ρg[2], RP=s.R.RP, Rx=s.R.Rx, Ry=s.R.Ry, Rz=s.R.Rz)
    for c in ("xx", "yy", "zz", "yz", "xz", "xy"):
        vals["t" + c], vals["to" + c], vals["e" + c] = getattr(s.τ, c), getattr(s.τ_o, c), getattr(s.ε, c)
    # centre copies of the shear stresses take part in multi_copy! only if someone materialised them
    cs = ("yz_c", "xz_c", "xy_c")
    if any(_opt(s.τ, c) is not None or _opt(s.τ_o, c) is not None for c in cs):
        for c in cs:
            vals["t" + c], vals["to" + c] = getattr(s.τ, c), getattr(s.τ_o, c)
    for n in _lib.F3_NAMES:
        setattr(f, n, ptr(vals.get(n)))
    f._keep = vals
    return f


def fields2d(stokes: StokesArrays, ρg, K, G) -> _lib.Stokes2DFields:
    f = _lib.Stokes2DFields()
    s = stokes
    vals = dict(P=s.P, P0=s.P0, divV=s.divV, Q=s.Q, Vx=s.V.Vx, Vy=s.V.Vy, Ux=s.U.Ux, Uy=s.U.Uy,
                eta=s.viscosity.η, K=K, G=G, fx=ρg[0], fy=ρg[1], RP=s.R.RP, Rx=s.R.Rx, Ry=s.R.Ry)
    for c in ("xx", "yy", "xy"):
        vals["t" + c], vals["to" + c], vals["e" + c] = getattr(s.τ, c), getattr(s.τ_o, c), getattr(s.ε, c)
    if _opt(s.τ, "xy_c") is not None or _opt(s.τ_o, "xy_c") is not None:
        vals["txy_c"], vals["toxy_c"] = s.τ.xy_c, s.τ_o.xy_c
    for n in _lib.F2_NAMES:
        setattr(f, n, ptr(vals.get(n)))
    f._keep = vals
    return f


def _ng(d):
    gg = global_grid()
    return gg.n_g(d) if gg.initialized else None


def _is_displacement(flow_bcs):
    from .arrays import DisplacementBoundaryConditions
    return isinstance(flow_bcs, DisplacementBoundaryConditions)


def params3d(stokes, pt, grid, flow_bcs, dt, *, iterMax=10_000, nout=500, b_width=(4, 4, 4), verbose=True, **_):
    ni = stokes._ni
    _di = _center_inv(grid)
    p = _lib.Stokes3DParams()
    p.nx, p.ny, p.nz = ni
    p.nxg, p.nyg, p.nzg = [(_ng(d) or ni[d]) for d in range(3)]
    p._dx, p._dy, p._dz = _di
    p.dt, p.r, p.theta_dtau, p.eta_dtau = float(dt), pt.r, pt.θ_dτ, pt.ηdτ
    p.eps_rel, p.eps_abs = pt.ϵ_rel, pt.ϵ_abs
    p.iterMax, p.nout = int(iterMax), int(nout)
    if flow_bcs is not None:
        p.free_slip, p.no_slip, p.periodic = (_lib.bcmask(flow_bcs.free_slip), _lib.bcmask(flow_bcs.no_slip),
                                              _lib.bcmask(flow_bcs.periodic))
    p.b_width[0], p.b_width[1], p.b_width[2] = [int(b) for b in b_width]
    p.verbose = int(bool(verbose))
    p.displacement_bcs = int(_is_displacement(flow_bcs))
    if getattr(grid, "nonuniform", False):
        raise NotImplementedError("non-uniform grids are built for the 2D drivers only")
    return p


def params2d(stokes, pt, grid, flow_bcs, dt, *, iterMax=10_000, nout=500, verbose=True, **_):
    ni = stokes._ni
    _di = _center_inv(grid)
    p = _lib.Stokes2DParams()
    p.nx, p.ny = ni
    p.nxg, p.nyg = [(_ng(d) or ni[d]) for d in range(2)]
    p._dx, p._dy = _di
    p.dt, p.r, p.theta_dtau, p.eta_dtau = float(dt), pt.r, pt.θ_dτ, pt.ηdτ
    p.eps_rel, p.eps_abs = pt.ϵ_rel, pt.ϵ_abs
    p.iterMax, p.nout = int(iterMax), int(nout)
    if flow_bcs is not None:
        p.free_slip, p.no_slip, p.periodic = (_lib.bcmask(flow_bcs.free_slip), _lib.bcmask(flow_bcs.no_slip),
                                              _lib.bcmask(flow_bcs.periodic))
    p.verbose = int(bool(verbose))
    p.displacement_bcs = int(_is_displacement(flow_bcs))
    _set_spacing2d(p, grid, stokes)
    return p


class _Hist:
    def __init__(self, cap):
        self.e1, self.e2 = np.zeros(cap), np.zeros(cap, dtype=np.int64)
        self.n = [np.zeros(cap) for _ in range(4)]
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        self.c = _lib.SolveResult(0, 0, cap, dp(self.e1), self.e2.ctypes.data_as(C.POINTER(C.c_int64)),
                                  dp(self.n[0]), dp(self.n[1]), dp(self.n[2]), dp(self.n[3]), 0.0, 0.0)

    def result(self, nD):
        k = self.c.nchecks
        r = SimpleNamespace(iter=self.c.iter, err_evo1=self.e1[:k].copy(), err_evo2=self.e2[:k].copy(),
                            norm_Rx=self.n[0][:k].copy(), norm_Ry=self.n[1][:k].copy(),
                            norm_divV=self.n[3][:k].copy(), time=self.c.time_s, av_time=self.c.av_time_s)
        if nD == 3:
            r.norm_Rz = self.n[2][:k].copy()
        setattr(r, "norm_∇V", r.norm_divV)
        return r


def rheology_table(phases) -> _lib.Rheology:
    """Per-phase material table for the C ABI.  `phases`: sequence of dict(eta, G, Kb[, C, phi_deg, psi_deg, eta_vp]) -- the
    LinearViscous / ConstantElasticity / DruckerPrager_regularised parameters of each GeoParams MaterialParams
    (C is GeoParams' cohesion parameter, i.e. τ_y / cos ϕ in test_shearband2D.jl) -- plus, optionally, `density` (ConstantDensity,
    PT_Density, T_Density, Compressible_Density) and `g` (gravity of the first phase), `softening_C` / `softening_phi`
    (LinearSoftening, NonLinearSoftening) and `creep` (Arrhenius law on top of eta): see jrx_rheology in include/jrx.h."""
    import math
    if isinstance(phases, _lib.Rheology):
        return phases
    r = _lib.Rheology()
    r.nphase = len(phases)
    if not 1 <= r.nphase <= _lib.MAXPHASE:
        raise ValueError(f"1..{_lib.MAXPHASE} phases supported")
    for q, ph in enumerate(phases):
        r.eta[q], r.G[q], r.Kb[q] = ph.get("eta", 0.0), ph["G"], ph["Kb"]
        _cr = ph.get("creep")
        if _cr is not None and _cr.get("kind") in ("dislocation", "powerlaw"):
            # GeoParams sums the strain rates of the elements of a CompositeRheology; the table holds ONE viscous element per phase, so a LinearViscous
            # element (eta) beside a DislocationCreep would be dropped silently: refuse it
            if r.eta[q] != 0.0:
                raise ValueError(f"phase {q}: `eta` (a LinearViscous element) in series with a dislocation creep is not built: one viscous element per phase")
        elif "eta" not in ph:
            raise KeyError("eta")
        pl = ph.get("C") is not None
        r.is_pl[q] = int(pl)
        if pl:
            r.C[q], r.eta_vp[q] = ph["C"], ph.get("eta_vp", 0.0)
            r.sinphi[q], r.cosphi[q] = math.sin(math.radians(ph["phi_deg"])), math.cos(math.radians(ph["phi_deg"]))
            r.sinpsi[q] = math.sin(math.radians(ph.get("psi_deg", 0.0)))
        # density / gravity (compute_ρg!): ph["density"] = dict(kind="constant"|"PT"|"T"|"compressible", rho0[, alpha, beta, T0, P0]); ph["g"]
        d = ph.get("density")
        if d is not None:
            r.has_density = 1
            r.rho_kind[q] = {"constant": 0, "PT": 1, "T": 2, "compressible": 3}[d.get("kind", "constant")]
            r.rho0[q], r.alpha[q], r.beta[q] = d["rho0"], d.get("alpha", 0.0), d.get("beta", 0.0)
            r.T0[q], r.P0[q] = d.get("T0", 0.0), d.get("P0", 0.0)
        if q == 0:
            r.gravity = float(ph.get("g", 0.0))
        # strain softening of C and ϕ: dict(kind="linear", min, max, lo, hi) | dict(kind="nonlinear", xi0, Delta[, mu=1, sigma=0.5])
        r.phi_deg[q] = float(ph.get("phi_deg", 0.0))
        for key, pre in (("softening_C", "softC_"), ("softening_phi", "softphi_")):
            sft = ph.get(key)
            if sft is None:
                continue
            if sft["kind"] == "linear":
                vals = (1, sft["min"], sft["max"], sft["lo"], sft["hi"])
            elif sft["kind"] == "nonlinear":
                vals = (2, sft["xi0"], sft["Delta"], sft.get("mu", 1.0), sft.get("sigma", 0.5))
            else:
                raise ValueError(f"unknown softening law {sft['kind']!r}")
            getattr(r, pre + "kind")[q] = vals[0]
            for name, v in zip("abcd", vals[1:]):
                getattr(r, pre + name)[q] = float(v)
        # creep law: dict(kind="arrhenius", Ea, Va, T0, R, cutoff=(lo, hi)) on top of eta (= η0)
        cr = ph.get("creep")
        if cr is not None and cr.get("kind") in ("dislocation", "powerlaw"):
            # DislocationCreep(A, n, E, V, R) with r = 0; apparatus = "AxialCompression" (FT = √3, FE = 2/√3) | "SimpleShear" (2, 2) | "Invariant" (1, 1)
            FT, FE = {"AxialCompression": (3.0 ** 0.5, 2.0 / 3.0 ** 0.5), "SimpleShear": (2.0, 2.0), "Invariant": (1.0, 1.0)}[cr.get("apparatus", "AxialCompression")]
            r.visc_kind[q] = 2
            r.creep_A[q], r.creep_n[q], r.creep_FT[q], r.creep_FE[q] = cr["A"], cr["n"], cr.get("FT", FT), cr.get("FE", FE)
            r.Ea[q], r.Va[q], r.Rgas[q] = cr.get("E", 0.0), cr.get("V", 0.0), cr.get("R", 8.3145)
        elif cr is not None:
            if cr.get("kind") != "arrhenius":
                raise ValueError(f"unknown creep law {cr.get('kind')!r}")
            r.visc_kind[q] = 1
            r.Ea[q], r.Va[q], r.Tref[q], r.Rgas[q] = cr["Ea"], cr["Va"], cr["T0"], cr.get("R", 8.3145)
            lo, hi = cr.get("cutoff", (0.0, float("inf")))
            r.visc_lo[q], r.visc_hi[q] = lo, hi
    return r


def _args_T(args):
    """args.T of the reference's `args` NamedTuple (dict or namespace here); None when absent"""
    if args is None:
        return None
    return args.get("T") if isinstance(args, dict) else getattr(args, "T", None)


def vep_fields2d(stokes, ρg, phase_ratios, args=None, allow_ghosted_T=False, strain_increment=False) -> _lib.VEP2DFields:
    s = stokes
    vals = dict(P=s.P, P0=s.P0, divV=s.divV, Q=s.Q, Vx=s.V.Vx, Vy=s.V.Vy, Ux=s.U.Ux, Uy=s.U.Uy,
                exx=s.ε.xx, eyy=s.ε.yy, exy=s.ε.xy, exy_c=s.ε.xy_c,
                eplxx=s.ε_pl.xx, eplyy=s.ε_pl.yy, eplxy=s.ε_pl.xy, eplxy_c=s.ε_pl.xy_c, dexy_c=s.Δε.xy_c, dexy=s.Δε.xy,
                txx=s.τ.xx, tyy=s.τ.yy, txy=s.τ.xy, txy_c=s.τ.xy_c, tII=s.τ.II,
                toxx=s.τ_o.xx, toyy=s.τ_o.yy, toxy=s.τ_o.xy, toxy_c=s.τ_o.xy_c,
                eta=s.viscosity.η, eta_v=s.viscosity.ηv, eta_vep=s.viscosity.η_vep,
                EII_pl=s.EII_pl, evol_pl=s.ε_vol_pl, EVol_pl=s.EVol_pl, fx=ρg[0], fy=ρg[1], RP=s.R.RP, Rx=s.R.Rx, Ry=s.R.Ry,
                omega_xy=s.ω.xy, phase_c=phase_ratios.center, phase_v=phase_ratios.vertex, T=_args_T(args))
    if strain_increment:      # Δε.xx, Δε.yy and ∇U are allocated on first use
        vals.update(dexx=s.Δε.xx, deyy=s.Δε.yy, divU=getattr(s, "∇U"))      # "∇" is not a Python identifier character
    if vals["T"] is not None and tuple(vals["T"].shape) != tuple(s._ni) and not allow_ghosted_T:
        raise ValueError(f"args.T must be cell-centred {tuple(s._ni)} (thermal.Tc), got {tuple(vals['T'].shape)}")
    f = _lib.VEP2DFields()
    for n in _lib.VEP_NAMES:
        setattr(f, n, ptr(vals.get(n)))
    f._keep = vals
    return f


def vep_params2d(stokes, pt, grid, flow_bcs, dt, *, iterMax=50.0e3, iterMin=1.0e2, nout=500, verbose=True, λ_relaxation=0.2,
                 viscosity_relaxation=1.0e-2, viscosity_cutoff=(-float("inf"), float("inf")), strain_increment=False,
                 free_surface=False, **_):
    ni = stokes._ni
    _di = _center_inv(grid)
    p = _lib.VEP2DParams()
    p.nx, p.ny = ni
    p.nxg, p.nyg = [(_ng(d) or ni[d]) for d in range(2)]
    p._dx, p._dy = _di
    p.dt, p.r, p.theta_dtau, p.eta_dtau, p.eps_rel, p.eps_abs = float(dt), pt.r, pt.θ_dτ, pt.ηdτ, pt.ϵ_rel, pt.ϵ_abs
    p.iterMax, p.iterMin, p.nout = int(iterMax), int(iterMin), int(nout)
    if flow_bcs is not None:
        p.free_slip, p.no_slip, p.periodic = (_lib.bcmask(flow_bcs.free_slip), _lib.bcmask(flow_bcs.no_slip), _lib.bcmask(flow_bcs.periodic))
    p.lambda_relaxation, p.viscosity_relaxation = float(λ_relaxation), float(viscosity_relaxation)
    p.cutoff_lo, p.cutoff_hi = float(viscosity_cutoff[0]), float(viscosity_cutoff[1])
    p.verbose = int(bool(verbose))
    p.free_surface = int(bool(free_surface))
    p.displacement_bcs = int(_is_displacement(flow_bcs))
    p.strain_increment = int(bool(strain_increment))
    _set_spacing2d(p, grid, stokes)
    return p


def _solve_vep2d(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, kw, h):
    """solve!(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, igg; kwargs) -- Stokes2D.jl:577-866"""
    if len(stokes._ni) == 3:
        if kw.get("strain_increment"):
            raise NotImplementedError("the reference's 3D VEP driver has no strain_increment variant (Stokes3D.jl:447-668)")
        return _solve_vep3d(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, kw, h)
    p = vep_params2d(stokes, pt_stokes, grid, flow_bcs, dt, **kw)
    f = vep_fields2d(stokes, ρg, phase_ratios, args, allow_ghosted_T=True, strain_increment=bool(p.strain_increment))
    T = _args_T(args)
    if T is not None and tuple(T.shape) != tuple(stokes._ni):      # args.T = thermal.T (ghosted): update_ρg! reads it at [i, j], as the reference does
        if tuple(T.shape) != tuple(n + 2 for n in stokes._ni):
            raise ValueError(f"args.T must be ni {tuple(stokes._ni)} (thermal.Tc) or ni .+ 2 (thermal.T), got {tuple(T.shape)}")
        p.T_ghosted = 1
    rh = rheology_table(rheology)
    hist = _Hist(int(p.iterMax // p.nout + 2))
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_stokes2d_vep_solve", C.byref(f), C.byref(rh), C.byref(p), C.byref(hist.c))
    return hist.result(2)


def _solve_nonlinear2d(stokes, pt_stokes, grid, flow_bcs, ρg, rheology, args, dt, kw, h):
    """solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology::MaterialParams, args, dt, igg; kwargs) -- Stokes2D.jl:345-557 (single phase:
    compute_τ_nonlinear! + center2vertex!).  `rheology` is one phase dict of the table; `args.T` may be thermal.T (ghosted)."""
    if len(stokes._ni) != 2:
        raise NotImplementedError("the single-phase MaterialParams variant is 2D only here; the reference's 3D one (Stokes3D.jl:206-445) does not "
                                  "run as written (it references an undefined phase_ratios, :258-259)")
    kw = dict(kw)
    kw.setdefault("iterMax", 10.0e3)
    kw.pop("iterMin", None)
    p = vep_params2d(stokes, pt_stokes, grid, flow_bcs, dt, **kw)
    none_pr = SimpleNamespace(center=None, vertex=None)
    f = vep_fields2d(stokes, ρg, none_pr, args, allow_ghosted_T=True)
    T = _args_T(args)
    if T is not None:
        if tuple(T.shape) == tuple(n + 2 for n in stokes._ni):
            p.T_ghosted = 1
        elif tuple(T.shape) != tuple(stokes._ni):
            raise ValueError(f"args.T must be thermal.T {tuple(n + 2 for n in stokes._ni)} or cell-centred {tuple(stokes._ni)}")
    rh = rheology_table([rheology] if isinstance(rheology, dict) else rheology)
    hist = _Hist(int(p.iterMax // p.nout + 2))
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_stokes2d_nonlinear_solve", C.byref(f), C.byref(rh), C.byref(p), C.byref(hist.c))
    return hist.result(2)


def vep_fields3d(stokes, ρg, phase_ratios, args=None) -> _lib.VEP3DFields:
    s = stokes
    vals = dict(P=s.P, P0=s.P0, divV=s.divV, Q=s.Q, Vx=s.V.Vx, Vy=s.V.Vy, Vz=s.V.Vz, Ux=s.U.Ux, Uy=s.U.Uy, Uz=s.U.Uz,
                eta=s.viscosity.η, eta_vep=s.viscosity.η_vep, EII_pl=s.EII_pl, evol_pl=s.ε_vol_pl, EVol_pl=s.EVol_pl,
                fx=ρg[0], fy=ρg[1], fz=ρg[2], RP=s.R.RP, Rx=s.R.Rx, Ry=s.R.Ry, Rz=s.R.Rz,
                omega_yz=s.ω.yz, omega_xz=s.ω.xz, omega_xy=s.ω.xy, tII=s.τ.II,
                phase_c=phase_ratios.center, phase_yz=phase_ratios.yz, phase_xz=phase_ratios.xz, phase_xy=phase_ratios.xy, T=_args_T(args))
    for pre, T in (("e", s.ε), ("epl", s.ε_pl), ("t", s.τ), ("to", s.τ_o)):
        for c in ("xx", "yy", "zz", "yz", "xz", "xy"):
            vals[pre + c] = getattr(T, c)
        for c in ("yz", "xz", "xy"):
            vals[pre + c + "_c"] = getattr(T, c + "_c")
    for c in ("yz", "xz", "xy"):
        vals["de" + c], vals["de" + c + "_c"] = getattr(s.Δε, c), getattr(s.Δε, c + "_c")
    f = _lib.VEP3DFields()
    for n in _lib.VEP3_NAMES:
        setattr(f, n, ptr(vals.get(n)))
    f._keep = vals
    return f


def vep_params3d(stokes, pt, grid, flow_bcs, dt, *, iterMax=10.0e3, nout=500, verbose=True, λ_relaxation=0.2, viscosity_relaxation=1.0e-2,
                 viscosity_cutoff=(-float("inf"), float("inf")), b_width=(4, 4, 4), **_):
    ni = stokes._ni
    if getattr(grid, "nonuniform", False):
        raise NotImplementedError("non-uniform grids are built for the 2D drivers only")
    _di = _center_inv(grid)
    p = _lib.VEP3DParams()
    p.nx, p.ny, p.nz = ni
    p.nxg, p.nyg, p.nzg = [(_ng(d) or ni[d]) for d in range(3)]
    p._dx, p._dy, p._dz = _di
    p.dt, p.r, p.theta_dtau, p.eta_dtau, p.eps_rel, p.eps_abs = float(dt), pt.r, pt.θ_dτ, pt.ηdτ, pt.ϵ_rel, pt.ϵ_abs
    p.iterMax, p.nout = int(iterMax), int(nout)
    if flow_bcs is not None:
        p.free_slip, p.no_slip, p.periodic = (_lib.bcmask(flow_bcs.free_slip), _lib.bcmask(flow_bcs.no_slip), _lib.bcmask(flow_bcs.periodic))
    p.lambda_relaxation, p.viscosity_relaxation = float(λ_relaxation), float(viscosity_relaxation)
    p.cutoff_lo, p.cutoff_hi = float(viscosity_cutoff[0]), float(viscosity_cutoff[1])
    p.verbose = int(bool(verbose))
    p.displacement_bcs = int(_is_displacement(flow_bcs))
    p.b_width[0], p.b_width[1], p.b_width[2] = [int(b) for b in b_width]
    return p


def _solve_vep3d(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, kw, h):
    """solve!(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, igg; kwargs) in 3D -- Stokes3D.jl:447-668"""
    p = vep_params3d(stokes, pt_stokes, grid, flow_bcs, dt, **kw)
    f = vep_fields3d(stokes, ρg, phase_ratios, args)
    T = _args_T(args)
    if T is not None and tuple(T.shape) != tuple(stokes._ni):      # args.T = thermal.T (ghosted), as the miniapps pass it: update_ρg! reads it at [i, j, k]
        if tuple(T.shape) != tuple(n + 2 for n in stokes._ni):
            raise ValueError(f"args.T must be ni {tuple(stokes._ni)} or ni .+ 2 (thermal.T), got {tuple(T.shape)}")
        p.T_ghosted = 1
    rh = rheology_table(rheology)
    hist = _Hist(int(p.iterMax // p.nout + 2))
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_stokes3d_vep_solve", C.byref(f), C.byref(rh), C.byref(p), C.byref(hist.c))
    return hist.result(3)


def compute_τ_nonlinear_(stokes, θ, λ, rheology, dt, pt_stokes, *, phase_ratios=None, handle=None):
    """compute_τ_nonlinear!(@tensor_center(τ), τ.II, @tensor(τ_o), @strain, @plastic_strain, EII_pl, P, θ, η, η_vep, λ,
    rheology, dt, θ_dτ, args) -- StressKernels.jl:266-307 (single phase: `phase_ratios=None`, rheology phase 1) or
    :310-351 (phases at the cell centres).  2D only."""
    _require_gpu(stokes)
    if len(stokes._ni) != 2:
        raise NotImplementedError("3D compute_τ_nonlinear! is not built")
    h = handle or _lib.default_handle(stokes.P.device.index)
    fake = SimpleNamespace(_di=dict(center=(1.0, 1.0)))
    p = vep_params2d(stokes, pt_stokes, fake, None, dt)
    pr = phase_ratios or SimpleNamespace(center=None, vertex=None)
    f = vep_fields2d(stokes, (stokes.P, stokes.P), pr)
    rh = rheology_table(rheology)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_compute_tau_nonlinear2d", C.byref(f), C.c_void_p(ptr(θ)), C.c_void_p(ptr(λ)), C.byref(rh), C.byref(p),
           C.c_int32(0 if phase_ratios is None else 1))


def center2vertex_(vertex, center, *more, handle=None):
    """center2vertex!(vertex, center) 2D -- Interpolations.jl:101-114; with six arrays the 3D form (Interpolations.jl:139-178)"""
    if more:
        from .gridops import center2vertex3d_
        return center2vertex3d_(vertex, center, *more, handle=handle)
    _require_gpu(vertex)
    if vertex.dim() != 2:
        raise TypeError("3D: center2vertex!(vertex_yz, vertex_xz, vertex_xy, center_yz, center_xz, center_xy) -- center2vertex3d_")
    h = handle or _lib.default_handle(vertex.device.index)
    torch.cuda.current_stream(vertex.device).synchronize()
    h.call("jrx_center2vertex2d", C.c_void_p(ptr(vertex)), C.c_void_p(ptr(center)), C.c_int64(center.shape[0]), C.c_int64(center.shape[1]))


def _vel_triplets(stokes):
    V, U = stokes.V, stokes.U
    three = len(stokes._ni) == 3
    vs = [V.Vx, V.Vy] + ([V.Vz] if three else [None])
    us = [U.Ux, U.Uy] + ([U.Uz] if three else [None])
    arr = lambda ts: (C.c_void_p * 3)(*[ptr(t) for t in ts])
    n = (C.c_int64 * 3)(*[(t.numel() if t is not None else 0) for t in vs])
    return arr(vs), arr(us), n


def velocity2displacement_(stokes, dt, *, handle=None):
    """velocity2displacement!(stokes, dt): U = V·dt -- types/displacement.jl:2-28"""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    v, u, n = _vel_triplets(stokes)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_velocity2displacement", u, v, n, C.c_double(float(dt)))


def displacement2velocity_(stokes, dt, bcs=None, *, handle=None):
    """displacement2velocity!(stokes, dt[, flow_bcs]): V = U·inv(dt); a no-op for VelocityBoundaryConditions -- types/displacement.jl:32-70"""
    from .arrays import DisplacementBoundaryConditions, VelocityBoundaryConditions
    if isinstance(bcs, VelocityBoundaryConditions):
        return
    if bcs is not None and not isinstance(bcs, DisplacementBoundaryConditions):
        raise TypeError(f"Unknown boundary conditions type: {type(bcs).__name__}")
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    v, u, n = _vel_triplets(stokes)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_displacement2velocity", v, u, n, C.c_double(float(dt)))


def compute_dt_(stokes, di, dt_diff=float("inf"), igg=None, *, handle=None):
    """compute_dt(stokes, di[, dt_diff][, igg]) = min(dt_diff, 0.9·min_d(di[d]/max|V_d|)) -- Utils.jl:492-519 (the maximum is taken over all
    ranks whenever the handle has a communicator, i.e. the `igg` forms)"""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    v, _, n = _vel_triplets(stokes)
    nd = len(stokes._ni)
    d = (C.c_double * 3)(*[float(x) for x in tuple(di)[:nd]], *([0.0] * (3 - nd)))
    out = C.c_double(0.0)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_compute_dt", v, n, d, C.c_int32(nd), C.c_double(float(dt_diff)), C.byref(out))
    return out.value


def shear2center_(A, *, handle=None):
    """shear2center!(A::SymmetricTensor) -- Interpolations.jl:291-323"""
    _require_gpu(A.xx)
    h = handle or _lib.default_handle(A.xx.device.index)
    torch.cuda.current_stream(A.xx.device).synchronize()
    n = [C.c_int64(m) for m in A.xx.shape]
    if A.xx.dim() == 3:
        h.call("jrx_shear2center3d", *[C.c_void_p(ptr(getattr(A, k))) for k in ("yz_c", "xz_c", "xy_c", "yz", "xz", "xy")], *n)
    else:
        h.call("jrx_shear2center2d", C.c_void_p(ptr(A.xy_c)), C.c_void_p(ptr(A.xy)), *n)


def accumulate_tensor_(II, A, dt, *, handle=None):
    """accumulate_tensor!(II, A::SymmetricTensor, dt): II += second_invariant_staggered(A) * dt -- StressKernels.jl:364-408"""
    _require_gpu(II)
    h = handle or _lib.default_handle(II.device.index)
    torch.cuda.current_stream(II.device).synchronize()
    n = [C.c_int64(m) for m in II.shape]
    comps = ("xx", "yy", "zz", "yz", "xz", "xy") if II.dim() == 3 else ("xx", "yy", "xy")
    h.call("jrx_accumulate_tensor3d" if II.dim() == 3 else "jrx_accumulate_tensor2d", C.c_void_p(ptr(II)),
           *[C.c_void_p(ptr(getattr(A, k))) for k in comps], C.c_double(float(dt)), *n)


def accumulate_vol_(EVol_pl, ε_vol_pl, dt, *, handle=None):
    """accumulate_vol!(EVol_pl, ε_vol_pl, dt) -- StressKernels.jl:410-431"""
    _require_gpu(EVol_pl)
    h = handle or _lib.default_handle(EVol_pl.device.index)
    torch.cuda.current_stream(EVol_pl.device).synchronize()
    h.call("jrx_accumulate_vol", C.c_void_p(ptr(EVol_pl)), C.c_void_p(ptr(ε_vol_pl)), C.c_double(float(dt)), C.c_int64(EVol_pl.numel()))


def compute_vorticity_(stokes, grid_or_di, *, handle=None):
    """compute_vorticity!(stokes.ω..., @velocity(stokes)..., _di) as the VEP drivers call it -- stress_rotation_particles.jl:17-50"""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    if getattr(grid_or_di, "nonuniform", False):
        raise NotImplementedError("compute_vorticity! alone takes a uniform grid (inside the 2D solves the spacing vectors are used)")
    _di = _center_inv(_as_grid(stokes, grid_or_di))
    torch.cuda.current_stream(stokes.P.device).synchronize()
    n = [C.c_int64(m) for m in stokes._ni]
    V = stokes.V
    if len(stokes._ni) == 3:
        h.call("jrx_compute_vorticity3d", *[C.c_void_p(ptr(x)) for x in (stokes.ω.yz, stokes.ω.xz, stokes.ω.xy, V.Vx, V.Vy, V.Vz)], *n,
               *[C.c_double(d) for d in _di])
    else:
        h.call("jrx_compute_vorticity2d", *[C.c_void_p(ptr(x)) for x in (stokes.ω.xy, V.Vx, V.Vy)], *n, *[C.c_double(d) for d in _di])


def tensor_invariant_(A, *, handle=None):
    """tensor_invariant!(A::SymmetricTensor) -- StressKernels.jl:443-487"""
    _require_gpu(A.xx)
    h = handle or _lib.default_handle(A.xx.device.index)
    torch.cuda.current_stream(A.xx.device).synchronize()
    if A.xx.dim() == 3:         # StressKernels.jl:472-487
        h.call("jrx_tensor_invariant3d", *[C.c_void_p(ptr(getattr(A, k))) for k in ("II", "xx", "yy", "zz", "yz", "xz", "xy")],
               *[C.c_int64(n) for n in A.xx.shape])
        return
    h.call("jrx_tensor_invariant2d", C.c_void_p(ptr(A.II)), C.c_void_p(ptr(A.xx)), C.c_void_p(ptr(A.yy)), C.c_void_p(ptr(A.xy)),
           C.c_int64(A.xx.shape[0]), C.c_int64(A.xx.shape[1]))


def _ghosted_T_flag(stokes, T):
    if T is None or tuple(T.shape) == tuple(stokes._ni):
        return 0
    if tuple(T.shape) != tuple(n + 2 for n in stokes._ni):
        raise ValueError(f"args.T must be ni {tuple(stokes._ni)} or ni .+ 2 (thermal.T), got {tuple(T.shape)}")
    return 1


def compute_viscosity_τII_(stokes, *rest, relaxation=1.0, handle=None):
    """compute_viscosity_τII!(stokes, phase_ratios, args, rheology, cutoff; relaxation) / update_viscosity_τII! (rheology/Viscosity.jl:67-106,198-216)"""
    return compute_viscosity_(stokes, *rest, relaxation=relaxation, handle=handle, fn="τII")


def compute_viscosity_(stokes, *rest, relaxation=1.0, handle=None, fn="εII", AII=None):
    """compute_viscosity!(stokes, phase_ratios, args, rheology, cutoff; relaxation) for the table rheology (rheology/Viscosity.jl:203-216), or -- without
    phase ratios -- compute_viscosity!(stokes, args, rheology::MaterialParams, cutoff; relaxation) (Viscosity.jl:118-167; args.T is thermal.T, read at I .+ 1).
    fn: "εII" (compute_viscosity!) or "τII" (compute_viscosity_τII! / update_viscosity_τII!): the invariant a power-law creep is evaluated at.
    AII (single-material form): the invariant array of compute_viscosity_εII! / _τII!(η, ν, AII, args, rheology, cutoff) (Viscosity.jl:169-196)"""
    if fn not in ("εII", "τII"):
        raise ValueError("fn must be 'εII' or 'τII'")
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    from .arrays import PhaseRatios
    if not rest or not (isinstance(rest[0], PhaseRatios) or hasattr(rest[0], "center")):
        args, rheology = rest[0], rest[1]
        cutoff = rest[2] if len(rest) > 2 else (-float("inf"), float("inf"))
        if isinstance(rheology, (list, tuple)):
            raise TypeError("several MaterialParams need phase ratios: compute_viscosity!(stokes, phase_ratios, args, rheology, cutoff)")
        rh = rheology_table([rheology])
        get = (lambda k: args.get(k)) if isinstance(args, dict) else (lambda k: getattr(args, k, None))
        T, P = (get("T"), get("P")) if args is not None else (None, None)
        η = stokes.viscosity.η
        nd = η.dim()
        n = (C.c_int64 * 3)(*η.shape, *([1] * (3 - nd)))
        td = (C.c_int64 * 3)(*(T.shape if T is not None else η.shape), *([1] * (3 - nd)))
        torch.cuda.current_stream(η.device).synchronize()
        h.call("jrx_compute_viscosity_single", C.c_void_p(ptr(η)), C.byref(rh), C.c_void_p(ptr(T)), C.c_void_p(ptr(P)), n, td, C.c_int32(nd),
               C.c_double(float(relaxation)), C.c_double(float(cutoff[0])), C.c_double(float(cutoff[1])), C.c_void_p(ptr(AII)), C.c_int32(int(fn == "τII")))
        return
    phase_ratios, args, rheology = rest[0], rest[1], rest[2]
    cutoff = rest[3] if len(rest) > 3 else (-float("inf"), float("inf"))
    sfx = "_tauII" if fn == "τII" else ""
    pt = SimpleNamespace(r=0.0, θ_dτ=1.0, ηdτ=1.0, ϵ_rel=0.0, ϵ_abs=0.0)
    if len(stokes._ni) == 3:
        fake = SimpleNamespace(_di=dict(center=(1.0, 1.0, 1.0)))
        p = vep_params3d(stokes, pt, fake, None, 1.0, viscosity_cutoff=cutoff)
        f = vep_fields3d(stokes, (stokes.P, stokes.P, stokes.P), phase_ratios, args)
        p.T_ghosted = _ghosted_T_flag(stokes, _args_T(args))
        torch.cuda.current_stream(stokes.P.device).synchronize()
        h.call("jrx_vep3d_compute_viscosity" + sfx, C.byref(f), C.byref(rheology_table(rheology)), C.byref(p), C.c_double(float(relaxation)))
        return
    fake = SimpleNamespace(_di=dict(center=(1.0, 1.0)))
    p = vep_params2d(stokes, pt, fake, None, 1.0, viscosity_cutoff=cutoff)
    f = vep_fields2d(stokes, (stokes.P, stokes.P), phase_ratios, args, allow_ghosted_T=True)
    p.T_ghosted = _ghosted_T_flag(stokes, _args_T(args))
    rh = rheology_table(rheology)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    h.call("jrx_vep2d_compute_viscosity" + sfx, C.byref(f), C.byref(rh), C.byref(p), C.c_double(float(relaxation)))


def solve_(stokes, pt_stokes, grid_or_di, flow_bcs, ρg, *rest, kwargs=None, handle=None):
    """solve!(stokes, pt_stokes, grid, flow_bcs, ρg, K, G, dt, igg; kwargs)   [3D, Stokes3D.jl:25-186]
       solve!(stokes, pt_stokes, grid, flow_bcs, ρg, G, K, dt, igg; kwargs)   [2D, Stokes2D.jl:181-325]
       solve!(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, igg; kwargs)   [2D VEP, Stokes2D.jl:577-866; 3D, Stokes3D.jl:447-668]
       solve!(stokes, pt_stokes, grid, flow_bcs, ρg, rheology, args, dt, igg; kwargs)   [2D single phase, Stokes2D.jl:345-557; rheology = one phase dict]

    `kwargs` is the reference's required keyword holding iterMax, nout, b_width, verbose.
    Returns the reference's NamedTuple as a namespace (iter, err_evo1, err_evo2, norm_Rx, ...).
    """
    _require_gpu(stokes)
    kw = dict(kwargs or {})
    h = handle or _lib.default_handle(stokes.P.device.index)
    grid = _as_grid(stokes, grid_or_di)
    ni, dev = stokes._ni, stokes.P.device
    nD = len(ni)
    from .arrays import PhaseRatios
    if rest and isinstance(rest[0], PhaseRatios):          # (phase_ratios, rheology, args, dt[, igg]) -> multiphase VEP variant
        phase_ratios, rheology, args, dt = rest[:4]
        return _solve_vep2d(stokes, pt_stokes, grid, flow_bcs, ρg, phase_ratios, rheology, args, dt, kw, h)
    if rest and isinstance(rest[0], dict):                   # (rheology::MaterialParams, args, dt[, igg]) -> single-phase VEP variant
        rheology, args, dt = rest[:3]
        return _solve_nonlinear2d(stokes, pt_stokes, grid, flow_bcs, ρg, rheology, args, dt, kw, h)
    if len(rest) < 3:
        raise TypeError("solve_: expected (K, G, dt[, igg]) / (G, K, dt[, igg]) or (phase_ratios, rheology, args, dt[, igg])")
    A, B, dt = rest[:3]
    if nD == 3:
        K, G = _as_field(A, ni, dev), _as_field(B, ni, dev)
        p = params3d(stokes, pt_stokes, grid, flow_bcs, dt, **kw)
        f = fields3d(stokes, ρg, K, G)
        hist = _Hist(int(p.iterMax // p.nout + 2))
        torch.cuda.current_stream(dev).synchronize()
        h.call("jrx_stokes3d_solve", C.byref(f), C.byref(p), C.byref(hist.c))
    else:
        G, K = _as_field(A, ni, dev), _as_field(B, ni, dev)
        p = params2d(stokes, pt_stokes, grid, flow_bcs, dt, **kw)
        f = fields2d(stokes, ρg, K, G)
        hist = _Hist(int(p.iterMax // p.nout + 2))
        torch.cuda.current_stream(dev).synchronize()
        h.call("jrx_stokes2d_solve", C.byref(f), C.byref(p), C.byref(hist.c))
    return hist.result(nD)


def flow_bcs_(stokes_or_V, bcs, *, handle=None):
    """flow_bcs!(stokes, bcs) -- BoundaryConditions.jl:65-100.  VelocityBoundaryConditions act on @velocity(stokes),
    DisplacementBoundaryConditions on @displacement(stokes) (BoundaryConditions.jl:71-78); the kernels are the same."""
    from .arrays import DisplacementBoundaryConditions
    if isinstance(bcs, DisplacementBoundaryConditions) and hasattr(stokes_or_V, "U"):
        U = stokes_or_V.U
        V = SimpleNamespace(Vx=U.Ux, Vy=U.Uy, Vz=getattr(U, "Uz", None))
    else:
        V = stokes_or_V.V if hasattr(stokes_or_V, "V") else stokes_or_V
    _require_gpu(V.Vx)
    h = handle or _lib.default_handle(V.Vx.device.index)
    torch.cuda.current_stream(V.Vx.device).synchronize()
    fs, ns, pe = _lib.bcmask(bcs.free_slip), _lib.bcmask(bcs.no_slip), _lib.bcmask(bcs.periodic)
    if hasattr(V, "Vz") and V.Vz is not None:
        nx, ny, nz = V.Vx.shape[0] - 1, V.Vy.shape[1] - 1, V.Vz.shape[2] - 1
        h.call("jrx_flow_bcs3d", C.c_void_p(ptr(V.Vx)), C.c_void_p(ptr(V.Vy)), C.c_void_p(ptr(V.Vz)),
               C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_uint32(fs), C.c_uint32(ns), C.c_uint32(pe))
    else:
        nx, ny = V.Vx.shape[0] - 1, V.Vy.shape[1] - 1
        h.call("jrx_flow_bcs2d", C.c_void_p(ptr(V.Vx)), C.c_void_p(ptr(V.Vy)), C.c_int64(nx), C.c_int64(ny),
               C.c_uint32(fs), C.c_uint32(ns), C.c_uint32(pe))


def compute_maxloc_(B, A, *, handle=None):
    """compute_maxloc!(B, A) with the default window (1,1[,1]) -- src/Utils.jl:409-461"""
    _require_gpu(A)
    h = handle or _lib.default_handle(A.device.index)
    torch.cuda.current_stream(A.device).synchronize()
    shp = list(A.shape) + [1] * (3 - A.dim())
    h.call("jrx_compute_maxloc", C.c_void_p(ptr(B)), C.c_void_p(ptr(A)), *[C.c_int64(n) for n in shp])
    return B


def sweep_stress_(stokes, pt, grid, K, G, dt, *, ητ=None, diag=True, handle=None):
    """compute_∇V! + compute_P! + compute_strain_rate! + compute_τ! fused (one stress sweep)."""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    ni = stokes._ni
    zero = [stokes.P] * len(ni)      # ρg is not read by this sweep
    if len(ni) == 3:
        f, p = fields3d(stokes, zero, K, G), params3d(stokes, pt, grid, None, dt)
        h.call("jrx_stokes3d_sweep_stress", C.byref(f), C.byref(p), C.c_int32(int(diag)))
    else:
        f, p = fields2d(stokes, zero, K, G), params2d(stokes, pt, grid, None, dt)
        h.call("jrx_stokes2d_sweep_stress", C.byref(f), C.c_void_p(ptr(ητ)), C.byref(p), C.c_int32(int(diag)))


def sweep_velocity_(stokes, pt, grid, ρg, ητ, dt, *, diag=True, handle=None):
    """compute_V! (+ velocity2displacement! when diag) -- one velocity sweep."""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    ni = stokes._ni
    if len(ni) == 3:
        f, p = fields3d(stokes, ρg, stokes.P, stokes.P), params3d(stokes, pt, grid, None, dt)
        h.call("jrx_stokes3d_sweep_velocity", C.byref(f), C.c_void_p(ptr(ητ)), C.byref(p), C.c_int32(int(diag)))
    else:
        f, p = fields2d(stokes, ρg, stokes.P, stokes.P), params2d(stokes, pt, grid, None, dt)
        h.call("jrx_stokes2d_sweep_velocity", C.byref(f), C.c_void_p(ptr(ητ)), C.byref(p), C.c_int32(int(diag)))


def compute_Res_(stokes, pt, grid, ρg, dt=1.0, *, handle=None):
    """compute_Res! (2D) -- VelocityKernels.jl:246-269"""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    f, p = fields2d(stokes, ρg, stokes.P, stokes.P), params2d(stokes, pt, grid, None, dt)
    h.call("jrx_stokes2d_compute_res", C.byref(f), C.byref(p))


def residual_sumsq(stokes, pt, grid, *, handle=None):
    """Σx² of the interior of Rx, Ry[, Rz] and of RP (local part of norm_mpi)."""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    ni = stokes._ni
    zero = [stokes.P] * len(ni)
    if len(ni) == 3:
        out = (C.c_double * 4)()
        f, p = fields3d(stokes, zero, stokes.P, stokes.P), params3d(stokes, pt, grid, None, 1.0)
        h.call("jrx_stokes3d_residual_sumsq", C.byref(f), C.byref(p), out)
    else:
        out = (C.c_double * 3)()
        f, p = fields2d(stokes, zero, stokes.P, stokes.P), params2d(stokes, pt, grid, None, 1.0)
        h.call("jrx_stokes2d_residual_sumsq", C.byref(f), C.byref(p), out)
    return np.array(out[:])


def iterate_timed_(stokes, pt, grid, flow_bcs, ρg, K, G, ητ, dt, iters, *, handle=None):
    """bench hook: `iters` PT iterations of the 3D loop body; returns
    (total_ms, stress_ms, velocity_ms, fused_group_ms, fused_kernel_ms, cells updated by the timed fused launch)."""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    f, p = fields3d(stokes, ρg, K, G), params3d(stokes, pt, grid, flow_bcs, dt)
    t = (C.c_double * 6)()
    h.call("jrx_stokes3d_iterate_timed", C.byref(f), C.c_void_p(ptr(ητ)), C.byref(p), C.c_int64(int(iters)), t)
    return tuple(t[:])
