"""Backend tags and traits -- mirror of the reference's dispatch surface.

Reference: src/JustRelax.jl:170-178 (CPUBackend / AMDGPUBackend tags, PTArray),
src/types/traits.jl:1-33 (BackendTrait family, backend(x)), ext/JustRelaxAMDGPUExt.jl:5-10.

This package implements exactly one backend: AMDGPUBackend (hand-written HIP for gfx950 behind
the C ABI of include/jrx.h).  The CPU backend of the reference is *not* re-implemented here;
asking for it raises, it never silently falls back.
"""
from __future__ import annotations

import torch


class CPUBackend:           # src/JustRelax.jl:170
    pass


class AMDGPUBackend:        # src/JustRelax.jl:176-178
    pass


class BackendTrait:         # src/types/traits.jl:1
    pass


class CPUBackendTrait(BackendTrait):
    pass


class NonCPUBackendTrait(BackendTrait):
    pass


class GPUBackendTrait(BackendTrait):
    pass


class AMDGPUBackendTrait(GPUBackendTrait):
    pass


def PTArray(backend_tag):
    """PTArray(::Type{Backend}) -> array constructor (ext/JustRelaxAMDGPUExt.jl:5-10)."""
    if backend_tag is AMDGPUBackend:
        return lambda a: torch.as_tensor(a, dtype=torch.float64).to(device_of(backend_tag))
    if backend_tag is CPUBackend:
        return lambda a: torch.as_tensor(a, dtype=torch.float64)
    raise ValueError(f"Backend {backend_tag} not supported")     # traits.jl:33 ArgumentError


def device_of(backend_tag) -> torch.device:
    if backend_tag is AMDGPUBackend:
        if not torch.cuda.is_available():
            raise RuntimeError("AMDGPUBackend requested but no HIP device is visible")
        return torch.device("cuda", torch.cuda.current_device())
    if backend_tag is CPUBackend:
        return torch.device("cpu")
    raise ValueError(f"Backend {backend_tag} not supported")


def backend(x) -> BackendTrait:
    """backend(x): trait of an array or of a struct of arrays (src/types/traits.jl:8-33)."""
    if isinstance(x, torch.Tensor):
        return AMDGPUBackendTrait() if x.is_cuda else CPUBackendTrait()
    for attr in ("P", "T"):          # backend(x::StokesArrays) = backend(x.P); ThermalArrays -> x.T
        if hasattr(x, attr) and isinstance(getattr(x, attr), torch.Tensor):
            return backend(getattr(x, attr))
    if hasattr(x, "_backend_array"):
        return backend(x._backend_array())
    raise ValueError(f"Backend of {type(x).__name__} not supported")
