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
    _di = grid._di if isinstance(grid, Geometry) else None
    return _di["center"] if isinstance(_di, dict) else _di


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


def solve_(stokes, pt_stokes, grid_or_di, flow_bcs, ρg, A, B, dt, igg=None, *, kwargs=None, handle=None):
    """solve!(stokes, pt_stokes, grid, flow_bcs, ρg, K, G, dt, igg; kwargs)   [3D, Stokes3D.jl:25-186]
       solve!(stokes, pt_stokes, grid, flow_bcs, ρg, G, K, dt, igg; kwargs)   [2D, Stokes2D.jl:181-325]

    `kwargs` is the reference's required keyword holding iterMax, nout, b_width, verbose.
    Returns the reference's NamedTuple as a namespace (iter, err_evo1, err_evo2, norm_Rx, ...).
    """
    _require_gpu(stokes)
    kw = dict(kwargs or {})
    h = handle or _lib.default_handle(stokes.P.device.index)
    grid = _as_grid(stokes, grid_or_di)
    ni, dev = stokes._ni, stokes.P.device
    nD = len(ni)
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
    """flow_bcs!(stokes, bcs) -- BoundaryConditions.jl:65-100"""
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
    """bench hook: `iters` PT iterations of the 3D loop body; returns (total_ms, stress_ms, velocity_ms, fused_ms)."""
    _require_gpu(stokes)
    h = handle or _lib.default_handle(stokes.P.device.index)
    torch.cuda.current_stream(stokes.P.device).synchronize()
    f, p = fields3d(stokes, ρg, K, G), params3d(stokes, pt, grid, flow_bcs, dt)
    t = (C.c_double * 4)()
    h.call("jrx_stokes3d_iterate_timed", C.byref(f), C.c_void_p(ptr(ητ)), C.byref(p), C.c_int64(int(iters)), t)
    return tuple(t[:])
