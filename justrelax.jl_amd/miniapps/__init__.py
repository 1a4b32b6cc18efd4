"""Synthetic-input builders that restate the reference's miniapps (the callers of the hot path).

Each builder returns a `Setup` holding host numpy arrays (Fortran order, named as in
include/jrx.h) plus the solver parameters, so that the same inputs can be fed to the HIP path
(via `upload`) and, in tests, to the CPU oracle.
"""
from .common import Setup, upload_stokes, download_stokes  # noqa: F401
from .stokes3d import solvi3d, taylor_green3d, random_fields3d, shearband3d, shearheating3d, vep_shapes3d, burstedde3d, burstedde_error_norms, plane_strain3d  # noqa: F401
from .stokes2d import (solcx2d, solkz2d, elastic_buildup2d, random_fields2d, shearband2d, sinking_block2d, shearheating2d,  # noqa: F401
                       thermal_convection2d)
from .thermal2d import diffusion2d, diffusion2d_multiphase, ball_phase_ratios  # noqa: F401
from .thermal3d import diffusion3d, diffusion3d_multiphase  # noqa: F401
