"""3D thermal diffusion input -- test/test_diffusion3D.jl:30-140."""
from __future__ import annotations

import math

import numpy as np

from ..arrays import TemperatureBoundaryConditions
from ..grid import Geometry, init_global_grid
from .common import Setup, fzeros_np
from .thermal2d import pt_thermal_coeffs_np


def thermal_shapes3d(nx, ny, nz):
    c, g = (nx, ny, nz), (nx + 2, ny + 2, nz + 2)
    s = {k: c for k in ("H", "shear_heating", "ResT", "K", "rhoCp", "thetar_dtau", "dtau_rho")}
    s.update(T=g, Told=g, dT=g, qTx=(nx + 1, ny, nz), qTx2=(nx + 1, ny, nz), qTy=(nx, ny + 1, nz), qTy2=(nx, ny + 1, nz),
             qTz=(nx, ny, nz + 1), qTz2=(nx, ny, nz + 1))
    return s


def diffusion3d(n=32, *, lx=100.0e3, ly=100.0e3, lz=100.0e3, ρ0=3.3e3, Cp0=1.2e3, K0=3.0, iterMax=50_000, nout=1000) -> Setup:
    """diffusion_3D (test_diffusion3D.jl:46-140): linear geotherm 1600..1900 K over z with a +100 K sphere of radius 10 km in the
    middle, H = 1e-6, constant K/Cp, PT_Density(ρ0 = 3.1e3, α = 1.5e-5); T = 300 K on top (k = end), 3500 K at the bottom, no
    flux on the sides; 10 steps of 50 kyr.  The ghost layers of T start at zero exactly as in the reference (init_T! fills
    k = 2..nz+1 only and the script applies no BC before the first solve)."""
    nx, ny, nz = (n, n, n) if isinstance(n, int) else tuple(n)
    kyr = 1.0e3 * 3600 * 24 * 365.25
    dt = 50 * kyr
    init_global_grid(nx, ny, nz)
    ni, li = (nx, ny, nz), (lx, ly, lz)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = Geometry(ni, li, origin=(0.0, 0.0, -lz))
    arr = {k: fzeros_np(s) for k, s in thermal_shapes3d(nx, ny, nz).items()}
    arr["H"][...] = 1.0e-6
    arr["K"][...] = K0
    arr["rhoCp"][...] = Cp0 * ρ0
    z = grid.xci[2]
    arr["T"][:, :, 1:-1] = (z * (1900.0 - 1600.0) / z.min() + 1600.0)[None, None, :]       # init_T! :30-33 over (1:nx+2, 1:ny+2, 1:nz)
    sides = dict(left=True, right=True, front=True, back=True)
    bc = TemperatureBoundaryConditions(no_flux=dict(sides, top=False, bot=False), constant_value=dict(sides, top=300.0, bot=3500.0))
    CFL = 0.95 / math.sqrt(3.1)
    th, dr = pt_thermal_coeffs_np(arr["K"], arr["rhoCp"], dt, di, li, CFL)
    arr["thetar_dtau"][...] = th
    arr["dtau_rho"][...] = dr
    rheology = dict(k=K0, Cp=Cp0, rho0=3.1e3, alpha=1.5e-5, T0=0.0)
    # elliptical_perturbation! :35-44
    X, Y, Z = np.meshgrid(*grid.xci, indexing="ij")
    arr["T"][1:-1, 1:-1, 1:-1][((X - lx / 2) ** 2 + (Y - ly / 2) ** 2 + (Z + lz / 2) ** 2) <= 10.0e3 ** 2] += 100.0
    return Setup(ni=ni, arrays=arr, grid=grid, pt=dict(eps=1.0e-8, CFL=CFL), dt=dt, flow_bcs=bc,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di, rheology=rheology, nt=10))


def diffusion3d_multiphase(n=32, *, iterMax=10_000, nout=100, sharp=False) -> Setup:
    """diffusion_3D of test/test_diffusion3D_multiphase.jl:83-207: the input of diffusion3d (H = 1e-6, ϵ = 1e-8, pt_thermal from the K / ρCp arrays)
    solved in the phase-ratio form with the two phases of the 2D multiphase test (ball = phase 2); 10 steps of 50 kyr, iterMax = 1e4, nout = 100."""
    from .thermal2d import MULTIPHASE_RHEOLOGY, ball_phase_ratios
    s = diffusion3d(n, iterMax=iterMax, nout=nout)
    lx, ly, lz = s.extra["li"]
    ph = ball_phase_ratios(s.grid, (lx / 2, ly / 2, -lz / 2), 10.0e3, s=1 if sharp else 4)
    s.arrays["P"] = np.zeros(s.ni, order="F")
    s.pt.update(max_lxyz=max(s.extra["li"]), Vpdtau=min(s.extra["di"]) * s.pt["CFL"])
    s.extra.update(rheology=MULTIPHASE_RHEOLOGY, phase_ratios=ph)
    return s
