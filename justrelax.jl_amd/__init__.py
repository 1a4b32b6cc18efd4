"""justrelax.jl_amd -- MI355X-native pseudo-transient Stokes / heat-diffusion hot path behind
JustRelax.jl's solve!(::StokesArrays, ...) / heatdiffusion_PT! surface.

The directory name contains a dot, so import it through `__graft_entry__.load_package()`
(registers it as the module `justrelax_jl_amd`).

Layout: csrc/ (hand-written HIP kernels for gfx950 + the C ABI declared in include/jrx.h),
arrays.py / backend.py / grid.py (host-side mirror of the reference's types and traits),
stokes.py / thermal.py / halo.py / gridops.py (the operator API: solve_, heatdiffusion_PT_, flow_bcs_, velocity2vertex_, ...),
miniapps/ (synthetic-input builders restating the reference's benchmark scripts).
Julia's `f!` is spelled `f_` here.
"""
from .backend import (AMDGPUBackend, AMDGPUBackendTrait, BackendTrait, CPUBackend, CPUBackendTrait,  # noqa: F401
                      GPUBackendTrait, NonCPUBackendTrait, PTArray, backend)
from .arrays import (DisplacementBoundaryConditions, PhaseRatios, PTStokesCoeffs, PTThermalCoeffs, StokesArrays,  # noqa: F401
                     SymmetricTensor, TemperatureBoundaryConditions, ThermalArrays, VelocityBoundaryConditions,
                     from_numpy, fzeros, to_numpy, trim_library_arrays, use_library_arrays)
from .grid import (IGG, Geometry, finalize_global_grid, init_global_grid, legacy_uniform_grid,  # noqa: F401
                   nx_g, ny_g, nz_g)
from .convert import Array_, PTArray_, checkpointing_npz, copy_, load_checkpoint_npz  # noqa: F401
from .vtk import pack_velocity, save_vtk  # noqa: F401
from . import miniapps  # noqa: F401


def __getattr__(name):
    # the operator API is imported lazily: it loads the HIP shared library and fails loudly if absent
    import importlib
    if name.startswith("_") or name in ("stokes", "thermal", "halo", "checks", "build", "arrays", "grid", "backend", "convert", "gridops"):
        raise AttributeError(name)
    for sub in ("stokes", "thermal", "halo", "gridops"):
        mod = importlib.import_module(f"{__name__}.{sub}")
        if hasattr(mod, name):
            return getattr(mod, name)
    raise AttributeError(name)
