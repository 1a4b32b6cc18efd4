"""2D thermal diffusion input -- test/test_diffusion2D.jl:27-125 (BASELINE config 1)."""
from __future__ import annotations

import math

import numpy as np

from ..arrays import TemperatureBoundaryConditions
from ..grid import Geometry, init_global_grid
from .common import Setup, fzeros_np


def thermal_shapes2d(nx, ny):
    c = (nx, ny)
    s = {k: c for k in ("H", "shear_heating", "ResT", "K", "rhoCp", "thetar_dtau", "dtau_rho")}
    s.update(T=(nx + 2, ny + 2), Told=(nx + 2, ny + 2), dT=(nx + 2, ny + 2),
             qTx=(nx + 1, ny), qTx2=(nx + 1, ny), qTy=(nx, ny + 1), qTy2=(nx, ny + 1))
    return s


def pt_thermal_coeffs_np(K, rhoCp, dt, di, li, CFL):
    """PTThermalCoeffs(K, ρCp, dt, di, li; CFL) on host arrays -- DiffusionPT_coefficients.jl:17-26"""
    Vpdτ = min(di) * CFL
    L = max(li)
    L2 = L ** 2
    Re = math.pi + np.sqrt(math.pi * math.pi + rhoCp * L2 / K / dt)
    return L / Vpdτ / Re, Vpdτ * L / K / Re


def diffusion2d(n=32, *, lx=100.0e3, ly=100.0e3, ρ0=3.3e3, Cp0=1.2e3, K0=3.0, iterMax=50_000, nout=1000) -> Setup:
    nx = ny = n
    kyr = 1.0e3 * 3600 * 24 * 365.25
    dt = 50 * kyr
    init_global_grid(nx, ny, 1)
    ni, li = (nx, ny), (lx, ly)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = Geometry(ni, li, origin=(0.0, -ly))
    arr = {k: fzeros_np(s) for k, s in thermal_shapes2d(nx, ny).items()}
    arr["H"][...] = 1.0e-6
    arr["K"][...] = K0
    arr["rhoCp"][...] = Cp0 * ρ0
    z = grid.xci[1]
    arr["T"][:, 1:-1] = (z * (1900.0 - 1600.0) / z.min() + 1600.0)[None, :]       # init_T! :27-30
    bc = TemperatureBoundaryConditions(no_flux=dict(left=True, right=True, top=False, bot=False),
                                       constant_value=dict(left=True, right=True, top=300.0, bot=3500.0))
    th, dr = pt_thermal_coeffs_np(arr["K"], arr["rhoCp"], dt, di, li, 0.95 / math.sqrt(2.1))
    arr["thetar_dtau"][...] = th
    arr["dtau_rho"][...] = dr
    rheology = dict(k=K0, Cp=Cp0, rho0=3.1e3, alpha=1.5e-5, T0=0.0)   # PT_Density(ρ0=3.1e3, β=0, T0=0, α=1.5e-5)
    return Setup(ni=ni, arrays=arr, grid=grid, pt=dict(eps=1.0e-8, CFL=0.95 / math.sqrt(2.1)), dt=dt, flow_bcs=bc,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False),
                 extra=dict(li=li, di=di, rheology=rheology, nt=int(math.ceil(1.0e3 * kyr / dt)),
                            perturbation=dict(δT=100.0, r=10.0e3, xc=lx / 2, yc=-ly / 2)))


def add_perturbation(T, grid, δT, r, xc, yc):
    """elliptical_perturbation! -- test_diffusion2D.jl:32-43 (cell-centre coordinates)."""
    X, Y = np.meshgrid(grid.xci[0], grid.xci[1], indexing="ij")
    T[1:-1, 1:-1][((X - xc) ** 2 + (Y - yc) ** 2) <= r ** 2] += δT
