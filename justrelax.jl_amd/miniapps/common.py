from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ..arrays import StokesArrays, from_numpy, to_numpy


@dataclass
class Setup:
    ni: tuple
    arrays: dict                       # name -> numpy F-ordered array
    grid: object = None                # Geometry
    pt: object = None                  # PTStokesCoeffs / PTThermalCoeffs inputs
    dt: float = float("inf")
    flow_bcs: object = None
    kwargs: dict = field(default_factory=dict)   # solve! kwargs (iterMax, nout, ...)
    extra: dict = field(default_factory=dict)


_STOKES_MAP3 = dict(P="P", P0="P0", divV="divV", Q="Q", RP="R.RP", Rx="R.Rx", Ry="R.Ry", Rz="R.Rz",
                    Vx="V.Vx", Vy="V.Vy", Vz="V.Vz", Ux="U.Ux", Uy="U.Uy", Uz="U.Uz", eta="viscosity.η")
for _c in ("xx", "yy", "zz", "yz", "xz", "xy"):
    _STOKES_MAP3["t" + _c] = "τ." + _c
    _STOKES_MAP3["to" + _c] = "τ_o." + _c
    _STOKES_MAP3["e" + _c] = "ε." + _c


def _get(obj, path):
    for p in path.split("."):
        obj = getattr(obj, p)
    return obj


def stokes_field_names(nd):
    names = dict(_STOKES_MAP3)
    if nd == 2:
        names = {k: v for k, v in names.items() if "z" not in k}
    return names


def upload_stokes(setup: Setup, backend_tag):
    """StokesArrays(backend, ni) filled from the host arrays; returns (stokes, ρg, K, G)."""
    from ..backend import device_of
    stokes = StokesArrays(backend_tag, setup.ni)
    dev = device_of(backend_tag)
    for name, path in stokes_field_names(len(setup.ni)).items():
        if name in setup.arrays:
            _get(stokes, path).copy_(from_numpy(setup.arrays[name], dev))
    rg = tuple(from_numpy(setup.arrays[f], dev) for f in ("fx", "fy", "fz")[: len(setup.ni)])
    K = from_numpy(setup.arrays["K"], dev)
    G = from_numpy(setup.arrays["G"], dev)
    return stokes, rg, K, G


def download_stokes(stokes) -> dict:
    out = {}
    for name, path in stokes_field_names(len(stokes._ni)).items():
        out[name] = to_numpy(_get(stokes, path))
    return out


def fzeros_np(shape):
    return np.zeros(shape, dtype=np.float64, order="F")


def stokes_shapes(ni) -> dict:
    """name -> extent for every Stokes array the C ABI takes (src/types/constructors/stokes.jl)."""
    from ..arrays import _tensor_shapes, residual_shapes, velocity_shapes
    ni = tuple(ni)
    s = {k: ni for k in ("P", "P0", "divV", "Q", "eta", "K", "G", "fx", "fy", "fz")[: None]}
    if len(ni) == 2:
        s.pop("fz")
    for k, shp in velocity_shapes(ni).items():
        s[k] = shp
        s[k.replace("V", "U")] = shp
    for k, shp in residual_shapes(ni).items():
        s[k] = shp
    for c, shp in _tensor_shapes(ni).items():
        if c.endswith("_v") or c == "II" or c.endswith("_c"):
            continue
        s["t" + c], s["to" + c], s["e" + c] = shp, shp, shp
    return s


def alloc_stokes(ni) -> dict:
    return {k: fzeros_np(v) for k, v in stokes_shapes(ni).items()}
