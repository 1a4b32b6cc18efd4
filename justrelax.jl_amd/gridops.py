"""Grid operators a time step runs either side of solve! / heatdiffusion_PT! -- the methods the reference's AMDGPU extension forwards to its
generic kernels (src/ext/AMDGPU/2D.jl:301-352, 3D.jl:311-362): velocity2vertex!, velocity2center!, vertex2center!, center2vertex! (3D),
center2vertex_harm!, compute_ρg!, compute_shear_heating!.  Julia's `f!` is spelled `f_`.  Every function forwards to one C-ABI entry point of
include/jrx.h (csrc/gridops.hip); nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .arrays import ptr
from .stokes import _require_gpu, rheology_table


def _h(t, handle):
    _require_gpu(t)
    h = handle or _lib.default_handle(t.device.index)
    torch.cuda.current_stream(t.device).synchronize()
    return h


def _p(*ts):
    return [C.c_void_p(ptr(t)) for t in ts]


def _i64(*ns):
    return [C.c_int64(int(n)) for n in ns]


def velocity2vertex_(*a, handle=None):
    """velocity2vertex!(Vx_v, Vy_v, Vx, Vy) / (Vx_v, Vy_v, Vz_v, Vx, Vy, Vz) -- Interpolations.jl:212-249; the kernel runs over size(Vx_v)"""
    if len(a) == 4:
        Vxv, Vyv, Vx, Vy = a
        if Vxv.shape != Vyv.shape:
            raise AssertionError("size(Vx_v) == size(Vy_v)")                                   # Interpolations.jl:238
        nx, ny = Vx.shape[0] - 1, Vx.shape[1] - 2
        if tuple(Vy.shape) != (nx + 2, ny + 1):
            raise ValueError("Vx (nx+1, ny+2) and Vy (nx+2, ny+1) do not belong to one grid")
        _h(Vxv, handle).call("jrx_velocity2vertex2d", *_p(Vxv, Vyv, Vx, Vy), *_i64(nx, ny, *Vxv.shape))
    elif len(a) == 6:
        Vxv, Vyv, Vzv, Vx, Vy, Vz = a
        if not (Vxv.shape == Vyv.shape == Vzv.shape):
            raise AssertionError("size(Vx_v) == size(Vy_v) == size(Vz_v)")                      # Interpolations.jl:213
        nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1] - 2, Vx.shape[2] - 2
        if tuple(Vy.shape) != (nx + 2, ny + 1, nz + 2) or tuple(Vz.shape) != (nx + 2, ny + 2, nz + 1):
            raise ValueError("Vx, Vy, Vz do not belong to one grid")
        _h(Vxv, handle).call("jrx_velocity2vertex3d", *_p(Vxv, Vyv, Vzv, Vx, Vy, Vz), *_i64(nx, ny, nz, *Vxv.shape))
    else:
        raise TypeError("velocity2vertex!(Vx_v, Vy_v, Vx, Vy) or (Vx_v, Vy_v, Vz_v, Vx, Vy, Vz)")


def velocity2vertex(Vx, Vy, Vz, *, handle=None):
    """velocity2vertex(Vx, Vy, Vz) -- Interpolations.jl:192-204 (allocating; outputs (nx, ny, nz) .- (1, 2, 2) of size(Vx), as the reference infers them)"""
    from .arrays import fzeros
    n = (Vx.shape[0] - 1, Vx.shape[1] - 2, Vx.shape[2] - 2)
    out = tuple(fzeros(n, Vx.device) for _ in range(3))
    velocity2vertex_(*out, Vx, Vy, Vz, handle=handle)
    return out


def velocity2center_(*a, handle=None):
    """velocity2center!(Vx_c, Vy_c, Vx, Vy) / (Vx_c, Vy_c, Vz_c, Vx, Vy, Vz) -- Interpolations.jl:257-289"""
    if len(a) == 4:
        Vxc, Vyc, Vx, Vy = a
        if Vxc.shape != Vyc.shape:
            raise AssertionError("size(Vx_c) == size(Vy_c)")
        nx, ny = Vx.shape[0] - 1, Vx.shape[1] - 2
        if tuple(Vxc.shape) != (nx, ny) or tuple(Vy.shape) != (nx + 2, ny + 1):
            raise ValueError("outputs must be (nx, ny) of the grid of Vx (nx+1, ny+2), Vy (nx+2, ny+1)")
        _h(Vxc, handle).call("jrx_velocity2center2d", *_p(Vxc, Vyc, Vx, Vy), *_i64(nx, ny))
    elif len(a) == 6:
        Vxc, Vyc, Vzc, Vx, Vy, Vz = a
        if not (Vxc.shape == Vyc.shape == Vzc.shape):
            raise AssertionError("size(Vx_c) == size(Vy_c) == size(Vz_c)")
        nx, ny, nz = Vx.shape[0] - 1, Vx.shape[1] - 2, Vx.shape[2] - 2
        if tuple(Vxc.shape) != (nx, ny, nz) or tuple(Vy.shape) != (nx + 2, ny + 1, nz + 2) or tuple(Vz.shape) != (nx + 2, ny + 2, nz + 1):
            raise ValueError("outputs must be ni of the grid of Vx, Vy, Vz")
        _h(Vxc, handle).call("jrx_velocity2center3d", *_p(Vxc, Vyc, Vzc, Vx, Vy, Vz), *_i64(nx, ny, nz))
    else:
        raise TypeError("velocity2center!(Vx_c, Vy_c, Vx, Vy) or (Vx_c, Vy_c, Vz_c, Vx, Vy, Vz)")


def vertex2center_(center, vertex, *, ghost_x=False, ghost_y=False, ghost_z=False, handle=None):
    """vertex2center!(center, vertex; ghost_x, ghost_y, ghost_z) -- Interpolations.jl:72-96"""
    nd = vertex.dim()
    if nd not in (2, 3) or center.dim() != nd:
        raise ValueError("2D or 3D arrays of the same rank")
    vd = (C.c_int64 * 3)(*vertex.shape, *([1] * (3 - nd)))
    cd = (C.c_int64 * 3)(*center.shape, *([1] * (3 - nd)))
    _h(center, handle).call("jrx_vertex2center", *_p(center, vertex), vd, cd, C.c_int32(nd), C.c_int32(bool(ghost_x)), C.c_int32(bool(ghost_y)),
                            C.c_int32(bool(ghost_z)))


def center2vertex_harm_(vertex, center, *, handle=None):
    """center2vertex_harm!(vertex, center) -- Interpolations.jl:116-137 (2D)"""
    nx, ny = center.shape
    if tuple(vertex.shape) != (nx + 1, ny + 1):
        raise ValueError("vertex must be size(center) .+ 1")
    _h(vertex, handle).call("jrx_center2vertex_harm2d", *_p(vertex, center), *_i64(nx, ny))


def center2vertex3d_(vertex_yz, vertex_xz, vertex_xy, center_yz, center_xz, center_xy, *, handle=None):
    """center2vertex!(vertex_yz, vertex_xz, vertex_xy, center_yz, center_xz, center_xy) -- Interpolations.jl:139-178"""
    nx, ny, nz = center_yz.shape
    want = ((nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz))
    if tuple(map(tuple, (vertex_yz.shape, vertex_xz.shape, vertex_xy.shape))) != want or not (center_yz.shape == center_xz.shape == center_xy.shape):
        raise ValueError("shear arrays (nx, ny+1, nz+1), (nx+1, ny, nz+1), (nx+1, ny+1, nz) and three centre arrays ni expected")
    _h(vertex_yz, handle).call("jrx_center2vertex3d", *_p(vertex_yz, vertex_xz, vertex_xy, center_yz, center_xz, center_xy), *_i64(nx, ny, nz))


def _args_get(args, k):
    if args is None:
        return None
    return args.get(k) if isinstance(args, dict) else getattr(args, k, None)


def compute_ρg_(ρg, *rest, handle=None):
    """compute_ρg!(ρg, rheology, args) / compute_ρg!(ρg, phase_ratios, rheology, args) -- rheology/BuoyancyForces.jl:6-60.  ρg: one array or the
    tuple of components (the scalar gravity of the first phase fills the last one, BuoyancyForces.jl:69-70); args: (; T, P) at the cell centres."""
    if len(rest) == 2:
        pr, (rheology, args) = None, rest
    elif len(rest) == 3:
        pr, rheology, args = rest
    else:
        raise TypeError("compute_ρg!(ρg, [phase_ratios,] rheology, args)")
    out = ρg[-1] if isinstance(ρg, (tuple, list)) else ρg
    rh = rheology_table([rheology] if isinstance(rheology, dict) else rheology)
    T, P = _args_get(args, "T"), _args_get(args, "P")
    nd = out.dim()
    if P is not None and tuple(P.shape) != tuple(out.shape):
        raise ValueError("args.P must have the shape of ρg (cell centres)")
    if T is not None and (T.dim() != nd or any(a < b for a, b in zip(T.shape, out.shape))):
        raise ValueError("args.T must be at least as large as ρg (a ghosted thermal.T is read at [i, j, k] without a shift, as in the reference)")
    pc = None
    if pr is not None:
        pc = pr.center
        if tuple(pc.shape) != (rh.nphase, *out.shape):
            raise ValueError("phase_ratios.center must be (nphase, ni...)")
    n = (C.c_int64 * 3)(*out.shape, *([1] * (3 - nd)))
    td = (C.c_int64 * 3)(*(T.shape if T is not None else out.shape), *([1] * (3 - nd)))
    _h(out, handle).call("jrx_compute_rhog", *_p(out), C.byref(rh), *_p(pc, T, P), n, td, C.c_int32(nd))


def compute_lithostatic_pressure_(P, ρg, dz, igg=None, *, handle=None):
    """compute_lithostatic_pressure!(P, ρg, dz[, igg]) -- src/Utils.jl:521-573: integrate ρg down the columns of the last dimension, P[j] = Σ_{k>j} ρg[k] dz[k] +
    ρg[j] dz[j] / 2; dz a number or one height per cell of that dimension.  With `igg` the result is the same as long as the vertical direction is not split
    across ranks (refused otherwise)."""
    if tuple(P.shape) != tuple(ρg.shape):
        raise ValueError(f"`P` and `ρg` must span the same cells, got {tuple(P.shape)} and {tuple(ρg.shape)}")          # DimensionMismatch, Utils.jl:562-566
    nd = P.dim()
    n = (C.c_int64 * 3)(*P.shape, *([1] * (3 - nd)))
    dzv = None
    if not isinstance(dz, (int, float)):
        import numpy as np
        host = np.ascontiguousarray(dz.detach().cpu().numpy() if isinstance(dz, torch.Tensor) else dz, dtype=np.float64)
        if host.shape != (P.shape[-1],):
            raise ValueError(f"`dz` must hold one height per cell, got {host.size} heights for {P.shape[-1]} cells")              # Utils.jl:612-616
        dzv = torch.tensor(host, dtype=torch.float64, device=P.device)
    _h(P, handle).call("jrx_compute_lithostatic_pressure", *_p(P, ρg), C.c_double(0.0 if dzv is not None else float(dz)), *_p(dzv), n, C.c_int32(nd))
    return P


def compute_shear_heating_(thermal, stokes, *rest, handle=None):
    """compute_shear_heating!(thermal, stokes, rheology, dt) / (thermal, stokes, phase_ratios, rheology, dt) -- thermal_diffusion/ShearHeating.jl:14-71.
    Each phase's table entry may carry `shear_heat` = Χ of its ConstantShearheating (absent: 0)."""
    if len(rest) == 2:
        pr, (rheology, dt) = None, rest
    elif len(rest) == 3:
        pr, rheology, dt = rest
    else:
        raise TypeError("compute_shear_heating!(thermal, stokes, [phase_ratios,] rheology, dt)")
    phases = [rheology] if isinstance(rheology, dict) else list(rheology)
    rh = rheology_table(phases)
    chi = (C.c_double * _lib.MAXPHASE)(*[float(p.get("shear_heat", 0.0)) for p in phases])
    sh = thermal.shear_heating
    nd = sh.dim()
    cen = ("xx", "yy", "zz", "yz_c", "xz_c", "xy_c") if nd == 3 else ("xx", "yy", "xy_c")
    stag = ("xx", "yy", "zz", "yz", "xz", "xy") if nd == 3 else ("xx", "yy", "xy")
    arr = lambda A, names: (C.c_void_p * 6)(*[ptr(getattr(A, k)) for k in names])
    n = (C.c_int64 * 3)(*sh.shape, *([1] * (3 - nd)))
    pc = pr.center if pr is not None else None
    _h(sh, handle).call("jrx_compute_shear_heating", *_p(sh), arr(stokes.τ, cen), arr(stokes.τ_o, cen), arr(stokes.ε, stag), *_p(pc), C.byref(rh), chi,
                        C.c_double(float(dt)), n, C.c_int32(nd))
