"""Field containers and PT coefficients -- host-side mirror of the reference's types.

Reference: src/types/stokes.jl:161-229, src/types/constructors/stokes.jl:10-303 (StokesArrays,
Velocity, SymmetricTensor, Residual, Viscosity, PTStokesCoeffs), src/types/heat_diffusion.jl,
src/types/constructors/heat_diffusion.jl:38-120 (ThermalArrays),
src/thermal_diffusion/DiffusionPT_coefficients.jl:17-26 (PTThermalCoeffs),
src/boundaryconditions/types.jl:65-181 (boundary-condition structs).

Data layout in HBM: every field is one dense fp64 array in Julia's column-major order (x fastest);
here that is a torch tensor whose *shape* is the Julia shape and whose strides are (1, n1, n1*n2),
so `t[i, j, k]` addresses the same element as Julia's `t[i+1, j+1, k+1]` and `t.data_ptr()` is
what `pointer(A)` would hand to the C ABI.  torch is used for device memory only.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

from .backend import AMDGPUBackend, CPUBackend, device_of


# ----------------------------------------------------------------------------- layout helpers
class _LibraryArray:
    """one array of jrx_field_alloc, exposed through __cuda_array_interface__ so that torch wraps it without a copy; torch keeps this object alive for as
    long as any tensor views the memory, and the last reference returns the array to the library"""

    def __init__(self, handle, count: int):
        import ctypes as C
        self._handle, p = handle, C.c_void_p()
        handle.call("jrx_field_alloc", C.c_int64(max(count, 1)), C.byref(p))
        self._ptr = p.value
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (self._ptr, False), "version": 2}

    def __del__(self):
        try:
            import ctypes as C
            if self._ptr and getattr(self._handle, "_h", None) and self._handle._h.value:      # a closed handle has released its arrays itself
                self._handle.lib.jrx_field_free(self._handle._h, C.c_void_p(self._ptr))
            self._ptr = 0
        except Exception:
            pass


_field_allocator = None      # a _lib.Handle: device arrays come from jrx_field_alloc (the backend owns the array constructor: src/ext/AMDGPU/3D.jl:46-48)


def use_library_arrays(handle):
    """Route the device arrays of every constructor of this package through jrx_field_alloc of `handle` (None: back to torch's allocator).  What
    the handle's option "field_placement" selects then holds for the caller's arrays as it does for the library's own."""
    global _field_allocator
    _field_allocator = handle


def trim_library_arrays(handle):
    """jrx_field_trim (include/jrx.h): the chunks of the placement pool ("field_placement" = 1) that no array took go back to the driver -- call it when the arrays of a run
    exist (the library's second state set is made by the first driver call)."""
    handle.call("jrx_field_trim")


def fzeros(shape, device, fill: float = 0.0) -> torch.Tensor:
    """Column-major fp64 array (the layout ParallelStencil's @zeros gives on every backend)."""
    shape = tuple(int(s) for s in shape)
    if _field_allocator is not None and torch.device(device).type == "cuda":
        count = 1
        for n_ in shape:
            count *= n_
        if count > 0:
            t = torch.as_tensor(_LibraryArray(_field_allocator, count), device=device)
            t.fill_(float(fill))
            return t.view(shape[::-1]).permute(*range(len(shape) - 1, -1, -1))
    t = torch.full(shape[::-1], float(fill), dtype=torch.float64, device=device)
    return t.permute(*range(len(shape) - 1, -1, -1))


def is_fortran(t: torch.Tensor) -> bool:
    exp, s = [], 1
    for n in t.shape:
        exp.append(s)
        s *= n
    return all(n == 1 or st == e for n, st, e in zip(t.shape, t.stride(), exp))


def from_numpy(a: np.ndarray, device) -> torch.Tensor:
    a = np.asarray(a, dtype=np.float64)
    src = torch.from_numpy(np.ascontiguousarray(a.T))
    if _field_allocator is not None and torch.device(device).type == "cuda" and a.size:
        t = fzeros(a.shape, device)
        t.permute(*range(a.ndim - 1, -1, -1)).copy_(src)
        return t
    t = src.to(device)
    return t.permute(*range(a.ndim - 1, -1, -1))


def to_numpy(t: torch.Tensor) -> np.ndarray:
    """Fortran-ordered numpy copy."""
    n = t.dim()
    c = t.permute(*range(n - 1, -1, -1)).contiguous().cpu().numpy()
    return c.T


def ptr(t) -> int:
    if t is None:
        return 0
    assert t.dtype == torch.float64 and is_fortran(t), "fields must be fp64 column-major"
    return t.data_ptr()


# ----------------------------------------------------------------------------- Stokes
def _tensor_shapes(ni):
    if len(ni) == 2:
        nx, ny = ni
        return dict(xx=ni, yy=ni, xx_v=(nx + 1, ny + 1), yy_v=(nx + 1, ny + 1), xy=(nx + 1, ny + 1), xy_c=ni, II=ni)
    nx, ny, nz = ni
    v = (nx + 1, ny + 1, nz + 1)
    return dict(xx=ni, yy=ni, zz=ni, xx_v=v, yy_v=v, zz_v=v, xy=(nx + 1, ny + 1, nz), yz=(nx, ny + 1, nz + 1),
                xz=(nx + 1, ny, nz + 1), yz_c=ni, xz_c=ni, xy_c=ni, II=ni)


class SymmetricTensor(SimpleNamespace):
    """src/types/constructors/stokes.jl:164-212"""

    def __init__(self, ni, device, lazy=()):
        super().__init__()
        self._ni, self._device = tuple(ni), device
        for k, shp in _tensor_shapes(ni).items():
            if k in lazy:
                continue
            setattr(self, k, fzeros(shp, device))

    def __getattr__(self, k):          # lazily allocate rarely used members on first touch
        shp = _tensor_shapes(self._ni).get(k)
        if shp is None:
            raise AttributeError(k)
        t = fzeros(shp, self._device)
        setattr(self, k, t)
        return t


def velocity_shapes(ni):
    """src/types/constructors/stokes.jl:10-34"""
    if len(ni) == 2:
        nx, ny = ni
        return dict(Vx=(nx + 1, ny + 2), Vy=(nx + 2, ny + 1))
    nx, ny, nz = ni
    return dict(Vx=(nx + 1, ny + 2, nz + 2), Vy=(nx + 2, ny + 1, nz + 2), Vz=(nx + 2, ny + 2, nz + 1))


def residual_shapes(ni):
    """src/types/constructors/stokes.jl:224-247"""
    if len(ni) == 2:
        nx, ny = ni
        return dict(RP=ni, Rx=(nx - 1, ny), Ry=(nx, ny - 1))
    nx, ny, nz = ni
    return dict(RP=ni, Rx=(nx - 1, ny, nz), Ry=(nx, ny - 1, nz), Rz=(nx, ny, nz - 1))


class _Lazy(SimpleNamespace):
    """namespace whose arrays are allocated on first access"""

    def __init__(self, device, shapes, fill=0.0):
        super().__init__()
        self.__dict__["_shapes"], self.__dict__["_device"], self.__dict__["_fill"] = dict(shapes), device, fill

    def __getattr__(self, k):
        shp = self.__dict__["_shapes"].get(k)
        if shp is None:
            raise AttributeError(k)
        t = fzeros(shp, self.__dict__["_device"], self.__dict__["_fill"])
        setattr(self, k, t)
        return t


class _Viscosity(_Lazy):
    """Viscosity(ni): η, η_vep, ητ (centres), ηv (vertices), all @ones -- constructors/stokes.jl:113-119"""

    def __init__(self, ni, device):
        super().__init__(device, dict(η_vep=ni, ητ=ni, ηv=tuple(n + 1 for n in ni)), fill=1.0)
        self.η = fzeros(ni, device, 1.0)


class PhaseRatios:
    """JustPIC.PhaseRatios(backend, nphases, ni): `center` (nphase, ni...) and `vertex` (nphase, ni.+1 ...) arrays, in 3D also
    `yz`, `xz`, `xy` on the edges, in CellArray layout (phase index fastest)."""

    def __init__(self, backend_tag, nphases, ni):
        dev = device_of(backend_tag)
        ni = tuple(int(n) for n in ni)
        self.nphases = int(nphases)
        self.center = fzeros((self.nphases,) + ni, dev)
        self.vertex = fzeros((self.nphases,) + tuple(n + 1 for n in ni), dev)
        # ratios at the velocity nodes (the heat-flux locations): Vx (nx+1, ny[, nz]), Vy (nx, ny+1[, nz]), Vz (nx, ny, nz+1)
        for d, name in enumerate(("Vx", "Vy", "Vz")[: len(ni)]):
            setattr(self, name, fzeros((self.nphases,) + tuple(n + (1 if e == d else 0) for e, n in enumerate(ni)), dev))
        if len(ni) == 3:               # phase ratios at the shear-stress locations (edges), 3D only
            ts = _tensor_shapes(ni)
            self.yz, self.xz, self.xy = (fzeros((self.nphases,) + ts[k], dev) for k in ("yz", "xz", "xy"))


class StokesArrays:
    """StokesArrays(backend, ni) -- src/types/constructors/stokes.jl:279-303.

    Members the pseudo-transient hot path never touches in the variants built here (vertex copies
    of the normal stresses, ε_pl, Δε, ω, λ, ...) are allocated lazily on first access, so that a
    512³ problem does not pay the reference's ~93 arrays (≈100 GB) up front.
    """

    _LAZY_T = ("xx_v", "yy_v", "zz_v", "II")

    def __init__(self, backend_tag, ni):
        if isinstance(ni, int) or any(not float(n).is_integer() for n in ni):
            raise ValueError("StokesArrays dimensions must be given as integers")   # types/stokes.jl:195-197
        ni = tuple(int(n) for n in ni)
        dev = device_of(backend_tag)
        self._device, self._ni = dev, ni
        self.P, self.P0, self.divV, self.Q = (fzeros(ni, dev) for _ in range(4))
        self.V = SimpleNamespace(**{k: fzeros(s, dev) for k, s in velocity_shapes(ni).items()})
        self.U = SimpleNamespace(**{k.replace("V", "U"): fzeros(s, dev) for k, s in velocity_shapes(ni).items()})
        self.τ = SymmetricTensor(ni, dev, lazy=self._LAZY_T)
        self.τ_o = SymmetricTensor(ni, dev, lazy=self._LAZY_T)
        self.ε = SymmetricTensor(ni, dev, lazy=self._LAZY_T + tuple(k for k in _tensor_shapes(ni) if k.endswith("_c")))
        self.viscosity = _Viscosity(ni, dev)                           # Viscosity: η = @ones (constructors/stokes.jl:113-119)
        # Vorticity (constructors/stokes.jl:79-100): 2D xy at the vertices; 3D yz, xz, xy at their staggered (edge) locations
        self.ω = _Lazy(dev, {"xy": tuple(n + 1 for n in ni)} if len(ni) == 2 else {k: _tensor_shapes(ni)[k] for k in ("yz", "xz", "xy")})
        self.R = SimpleNamespace(**{k: fzeros(s, dev) for k, s in residual_shapes(ni).items()})

    # ASCII aliases
    tau = property(lambda s: s.τ)
    tau_o = property(lambda s: s.τ_o)
    eps = property(lambda s: s.ε)

    def __getattr__(self, k):          # lazily allocated, never used by the variants implemented here
        ni, dev = self.__dict__["_ni"], self.__dict__["_device"]
        lazy_center = ("EII_pl", "EVol_pl", "ε_vol_pl", "∇U", "λ", "ΔPψ")
        if k in lazy_center:
            t = fzeros(ni, dev)
        elif k == "λv":
            t = fzeros(tuple(n + 1 for n in ni), dev)
        elif k in ("ε_pl", "Δε"):
            t = SymmetricTensor(ni, dev, lazy=tuple(_tensor_shapes(ni)))
        elif k in ("η_vep", "ητ"):
            t = fzeros(ni, dev)
        else:
            raise AttributeError(k)
        setattr(self, k, t)
        return t


class PTStokesCoeffs:
    """src/types/stokes.jl:203-229"""

    def __init__(self, li, di, *, ϵ_rel=1.0e-6, ϵ_abs=1.0e-12, Re=3 * math.pi, CFL=None, r=0.7,
                 eps_rel=None, eps_abs=None):
        N = len(li)
        if CFL is None:
            CFL = 0.9 / math.sqrt(2.1) if N == 2 else 0.9 / math.sqrt(3.1)
        if eps_rel is not None:
            ϵ_rel = eps_rel
        if eps_abs is not None:
            ϵ_abs = eps_abs
        lτ = min(li)
        Vpdτ = min(di) * CFL
        self.CFL, self.ϵ_rel, self.ϵ_abs, self.Re, self.r = float(CFL), float(ϵ_rel), float(ϵ_abs), float(Re), float(r)
        self.Vpdτ = Vpdτ
        self.θ_dτ = lτ * (r + 4 / 3) / (Re * Vpdτ)
        self.ηdτ = Vpdτ * lτ / Re

    eps_rel = property(lambda s: s.ϵ_rel)
    eps_abs = property(lambda s: s.ϵ_abs)
    theta_dtau = property(lambda s: s.θ_dτ)
    eta_dtau = property(lambda s: s.ηdτ)


# ----------------------------------------------------------------------------- thermal
class ThermalArrays:
    """ThermalArrays(backend, ni) -- src/types/constructors/heat_diffusion.jl:38-120"""

    def __init__(self, backend_tag, ni):
        ni = tuple(int(n) for n in ni)
        dev = device_of(backend_tag)
        g = tuple(n + 2 for n in ni)
        self._ni, self._device = ni, dev
        self.T, self.Told, self.ΔT = fzeros(g, dev), fzeros(g, dev), fzeros(g, dev)
        self.adiabatic, self.dT_dt = fzeros(ni, dev), fzeros(ni, dev)
        names = "xyz"[: len(ni)]
        for d, c in enumerate(names):
            shp = tuple(n + (1 if e == d else 0) for e, n in enumerate(ni))
            setattr(self, f"qT{c}", fzeros(shp, dev))
            setattr(self, f"qT{c}2", fzeros(shp, dev))
        self.H, self.shear_heating, self.ResT = fzeros(ni, dev), fzeros(ni, dev), fzeros(ni, dev)


class PTThermalCoeffs:
    """PTThermalCoeffs(backend, K, ρCp, dt, di, li; ϵ, CFL) -- DiffusionPT_coefficients.jl:17-26.

    Re = π + √(π² + ρCp·L²/K/dt); θr_dτ = L/Vpdτ/Re; dτ_ρ = Vpdτ·L/K/Re (elementwise, in that order).
    """

    def __init__(self, backend_tag, K, ρCp, dt, di, li, *, ϵ=1.0e-8, CFL=0.9 / math.sqrt(3)):
        if isinstance(K, (list, tuple)):
            raise TypeError("multi-phase rheology: use PTThermalCoeffs.from_phases(backend, rheology, phase_ratios, args, dt, ni, di, li)")
        Vpdτ = min(di) * CFL
        max_lxyz = max(li)
        max_lxyz2 = max_lxyz ** 2
        Re = math.pi + torch.sqrt(math.pi * math.pi + ρCp * max_lxyz2 / K / dt)
        self.CFL, self.ϵ, self.max_lxyz, self.max_lxyz2, self.Vpdτ = CFL, ϵ, max_lxyz, max_lxyz2, Vpdτ
        self.θr_dτ = max_lxyz / Vpdτ / Re
        self.dτ_ρ = Vpdτ * max_lxyz / K / Re

    @classmethod
    def from_phases(cls, backend_tag, rheology, phase_ratios, args, dt, ni, di, li, *, ϵ=1.0e-8, CFL=0.9 / math.sqrt(3)):
        """PTThermalCoeffs(backend, rheology, phase_ratios, args, dt, ni, di, li; ϵ, CFL) -- DiffusionPT_coefficients.jl:39-66:
        the arrays are filled on the device by update_pt_thermal_arrays! (jrx_update_pt_thermal_arrays)."""
        from .thermal import update_pt_thermal_arrays_
        self = cls.__new__(cls)
        dev = device_of(backend_tag)
        self.CFL, self.ϵ, self.max_lxyz, self.max_lxyz2, self.Vpdτ = CFL, ϵ, max(li), max(li) ** 2, min(di) * CFL
        self.θr_dτ, self.dτ_ρ = fzeros(tuple(ni), dev), fzeros(tuple(ni), dev)
        update_pt_thermal_arrays_(self, phase_ratios, rheology, args, 1.0 / dt)
        return self


# ----------------------------------------------------------------------------- boundary conditions
_FACES2, _FACES3 = ("left", "right", "top", "bot"), ("left", "right", "front", "back", "top", "bot")


def _pairs(nD):
    return (("left", "right"), ("bot", "top")) if nD == 2 else (("left", "right"), ("front", "back"), ("bot", "top"))


def check_periodic_pairs(periodic, nD):
    """src/boundaryconditions/types.jl:183-195"""
    for a, b in _pairs(nD):
        if bool(periodic.get(a, False)) != bool(periodic.get(b, False)):
            raise ValueError(f"Periodic boundary conditions must be paired: {a} and {b}")


class _FlowBCs:
    """src/boundaryconditions/types.jl:108-181 (check_flow_bcs)"""

    def __init__(self, *, no_slip=None, free_slip=None, periodic=None, free_surface=False):
        given = [d for d in (no_slip, free_slip, periodic) if d is not None]
        nfaces = max((len(d) for d in given), default=4)
        faces = _FACES2 if nfaces == 4 else _FACES3
        no_slip = no_slip if no_slip is not None else {f: False for f in faces}
        free_slip = free_slip if free_slip is not None else {f: True for f in faces}
        periodic = periodic if periodic is not None else {f: False for f in faces}
        if not (len(no_slip) == len(free_slip) == len(periodic)):
            raise AssertionError("no_slip, free_slip and periodic must name the same faces")
        self.nD = 2 if nfaces == 4 else 3
        check_periodic_pairs(periodic, self.nD)
        for f in faces:
            if sum(bool(d.get(f, False)) for d in (no_slip, free_slip, periodic)) > 1:
                raise ValueError(f"Incompatible boundary conditions on the {f} boundary")
        if free_surface and periodic.get("top", False):
            raise ValueError("Incompatible boundary conditions: the top can't be both periodic and free_surface")
        self.no_slip, self.free_slip, self.periodic, self.free_surface = dict(no_slip), dict(free_slip), dict(periodic), free_surface


class VelocityBoundaryConditions(_FlowBCs):
    pass


class DisplacementBoundaryConditions(_FlowBCs):
    pass


class TemperatureBoundaryConditions:
    """src/boundaryconditions/types.jl:65-106"""

    def __init__(self, *, no_flux=None, constant_flux=None, constant_value=None, periodic=None, dirichlet=None):
        # dirichlet = dict(constant=number | None, mask=array of the shape of thermal.T | None) -- Dirichlet(constant, mask), Dirichlet.jl:131-135
        self.dirichlet = dict(dirichlet) if dirichlet else dict(constant=None, mask=None)
        d2 = dict(left=False, right=False, top=False, bot=False)
        no_flux = no_flux if no_flux is not None else dict(d2, left=True)
        given = [no_flux] + [d for d in (constant_flux, constant_value, periodic) if d is not None]
        self.nD = 2 if max(len(d) for d in given) == 4 else 3
        full = dict(front=False, back=False, **d2)
        self.no_flux = {**full, **no_flux}
        self.constant_flux = {**full, **(constant_flux or {})}
        self.constant_value = {**full, **(constant_value or {})}
        self.periodic = {**full, **(periodic or {})}
        check_periodic_pairs(self.periodic, self.nD)
        for k, v in self.periodic.items():
            if v and any(c[k] is not False for c in (self.no_flux, self.constant_flux, self.constant_value)):
                raise ValueError(f"Incompatible boundary conditions on the {k} boundary")


__all__ = ["PhaseRatios", "StokesArrays", "PTStokesCoeffs", "ThermalArrays", "PTThermalCoeffs", "SymmetricTensor",
           "VelocityBoundaryConditions", "DisplacementBoundaryConditions", "TemperatureBoundaryConditions",
           "fzeros", "from_numpy", "to_numpy", "ptr", "is_fortran", "AMDGPUBackend", "CPUBackend"]
