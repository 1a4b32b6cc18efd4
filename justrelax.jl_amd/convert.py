"""Conversions between device and host copies of the field containers, and checkpoint / restart on top of them.

Reference: src/types/type_conversions.jl:17-68 (`Array(stokes)`, `copy(stokes)`, `PTArray(backend, stokes)`) and the checkpoint functions built on
them, src/IO/JLD2.jl:37-64,125-149 (`checkpointing_jld2(dst, stokes[, thermal], time, timestep[, igg]; kwargs...)`, one file per rank named
`checkpointNNNN`, written to a temporary directory and moved into place; `load_checkpoint_jld2`).  JLD2 / HDF5 are not available here, so the
container format is numpy's `.npz` (one entry per array, keys are the field paths, e.g. "stokes/τ/xy"); the values are the Fortran-ordered arrays a
Julia reader would see.  This is host-side I/O around the hot path, not part of it.
"""
from __future__ import annotations

import os
import shutil
import tempfile
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from .arrays import StokesArrays, ThermalArrays, from_numpy, to_numpy
from .backend import device_of


def _children(x):
    """(name, value) pairs of a container: public attributes that are tensors or containers; lazily allocated members only once they exist"""
    for k, v in vars(x).items():
        if k.startswith("_"):
            continue
        if isinstance(v, (torch.Tensor, np.ndarray)) or hasattr(v, "__dict__"):
            yield k, v


def Array_(x):
    """Array(x): host copy of a StokesArrays / ThermalArrays / SymmetricTensor / Velocity ... -- a tree of Fortran-ordered numpy arrays
    (type_conversions.jl:17-31).  None stays None; host arrays are returned as they are."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return to_numpy(x)
    if isinstance(x, np.ndarray):
        return x
    out = SimpleNamespace(**{k: Array_(v) for k, v in _children(x)})
    for meta in ("_ni",):
        if hasattr(x, meta):
            setattr(out, meta, getattr(x, meta))
    return out


def copy_(x):
    """copy(x): deep copy on the same device (type_conversions.jl:37-47)"""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        t = torch.empty_like(x)          # keeps the column-major strides
        t.copy_(x)
        return t
    if isinstance(x, np.ndarray):
        return x.copy(order="F")
    if isinstance(x, (StokesArrays, ThermalArrays)):
        y = type(x)(_backend_of(x), x._ni)
        _assign(y, x, lambda dst, src: dst.copy_(src))
        return y
    return SimpleNamespace(**{k: copy_(v) for k, v in _children(x)})


def _backend_of(x):
    from .backend import AMDGPUBackend, CPUBackend
    dev = getattr(x, "_device", None)
    return AMDGPUBackend if (dev is not None and dev.type == "cuda") else CPUBackend


def _assign(dst, src, put):
    """walk `src` and write every array into the member of the same path of `dst` (allocating lazy members of dst on first touch)"""
    for k, v in _children(src):
        if isinstance(v, (torch.Tensor, np.ndarray)):
            put(getattr(dst, k), v)
        else:
            _assign(getattr(dst, k), v, put)


def PTArray_(backend_tag, x, kind=None):
    """PTArray(backend, x): the container on `backend` with the values of the host (or device) tree x (type_conversions.jl:50-68) -- e.g. a
    checkpoint read back for a restart.  `kind`: StokesArrays or ThermalArrays (default: by the members of x)."""
    if x is None:
        return None
    dev = device_of(backend_tag)
    if isinstance(x, np.ndarray):
        return from_numpy(x, dev)
    if isinstance(x, torch.Tensor):
        return copy_(x).to(dev) if x.device != dev else copy_(x)
    kind = kind or (ThermalArrays if hasattr(x, "Told") else StokesArrays)
    y = kind(backend_tag, x._ni)
    _assign(y, x, lambda dst, src: dst.copy_(from_numpy(src, dev) if isinstance(src, np.ndarray) else src))
    return y


# ---------------------------------------------------------------------------------------------- checkpoint / restart
def checkpoint_name(dst, igg=None) -> str:
    """JLD2.jl:37-38 with the container's extension"""
    return f"{dst}/checkpoint.npz" if igg is None else f"{dst}/checkpoint{int(igg.me):04d}.npz"


def _flatten(prefix, tree, out):
    for k, v in vars(tree).items():
        if k.startswith("_"):
            continue
        if isinstance(v, np.ndarray):
            out[f"{prefix}/{k}"] = v
        elif v is not None:
            _flatten(f"{prefix}/{k}", v, out)


def checkpointing_npz(dst, stokes, thermal=None, time=0.0, timestep=0.0, igg=None, **kwargs):
    """checkpointing_jld2(dst, stokes[, thermal], time, timestep[, igg]; kwargs...) -- JLD2.jl:40-99: host copies of the containers, the scalars and any
    extra arrays given by keyword, written to a temporary directory first and moved over the previous checkpoint."""
    fname = checkpoint_name(dst, igg)
    Path(dst).mkdir(parents=True, exist_ok=True)
    items = {"time": np.float64(time), "timestep": np.float64(timestep)}
    hs = Array_(stokes)
    items["stokes/_ni"] = np.asarray(hs._ni, dtype=np.int64)
    _flatten("stokes", hs, items)
    if thermal is not None:
        ht = Array_(thermal)
        items["thermal/_ni"] = np.asarray(ht._ni, dtype=np.int64)
        _flatten("thermal", ht, items)
    for k, v in kwargs.items():
        if v is None:
            continue
        if isinstance(v, (tuple, list)):
            for q, a in enumerate(v):
                items[f"extra/{k}/{q}"] = Array_(a)
        else:
            items[f"extra/{k}"] = Array_(v) if isinstance(v, (torch.Tensor, np.ndarray)) else np.asarray(v)
    with tempfile.TemporaryDirectory() as tmp:
        tmpf = os.path.join(tmp, os.path.basename(fname))
        with open(tmpf, "wb") as fh:
            np.savez(fh, **items)
        shutil.move(tmpf, fname)


def _unflatten(items, prefix):
    root = SimpleNamespace()
    found = False
    for key, v in items.items():
        if not key.startswith(prefix + "/"):
            continue
        found = True
        parts = key[len(prefix) + 1:].split("/")
        node = root
        for p in parts[:-1]:
            if not hasattr(node, p):
                setattr(node, p, SimpleNamespace())
            node = getattr(node, p)
        if parts[-1] == "_ni":
            root._ni = tuple(int(n) for n in v)
        else:
            setattr(node, parts[-1], np.asfortranarray(v))
    return root if found else None


def load_checkpoint_npz(file_path, igg=None):
    """load_checkpoint_jld2(file_path[, igg]) -> (stokes, thermal | None, time, timestep) as host trees; PTArray_(backend, tree) puts them on a backend"""
    with np.load(checkpoint_name(file_path, igg)) as z:
        items = {k: z[k] for k in z.files}
    return _unflatten(items, "stokes"), _unflatten(items, "thermal"), float(items["time"]), float(items["timestep"])
