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


def _inside_fraction(lo, hi, centre, r, s):
    """Fraction of the box [lo, hi] (arrays per dimension, broadcast against each other) inside the ball |x - centre| <= r, by s^d midpoint
    samples -- stands in for the expectation of JustPIC's particle-count phase ratios (the reference seeds particles at random)."""
    nd = len(lo)
    frac = 0.0
    offs = (np.arange(s) + 0.5) / s
    for idx in np.ndindex(*(s,) * nd):
        d2 = 0.0
        for d in range(nd):
            x = lo[d] + (hi[d] - lo[d]) * offs[idx[d]]
            d2 = d2 + (x - centre[d]) ** 2
        frac = frac + (d2 <= r * r)
    return frac / s ** nd


def ball_phase_ratios(grid, centre, r, s=8):
    """Two-phase ratios of a ball (phase 2) in a matrix (phase 1) at the cell centres and at the velocity nodes (control volumes of one
    cell size around each node, clipped to the domain): dict(center, Vx, Vy[, Vz]) of arrays (2, shape...), phase index first (fastest)."""
    nd = len(grid.xvi)
    xv = [np.asarray(v) for v in grid.xvi]
    lo_c, hi_c = [v[:-1] for v in xv], [v[1:] for v in xv]

    def shaped(vals, d):
        sh = [1] * nd
        sh[d] = -1
        return vals.reshape(sh)

    def ratios(lo, hi):
        f = _inside_fraction([shaped(a, d) for d, a in enumerate(lo)], [shaped(a, d) for d, a in enumerate(hi)], centre, r, s)
        f = np.broadcast_to(f, tuple(len(a) for a in lo)).astype(np.float64)
        return np.asfortranarray(np.stack([1.0 - f, f], axis=0))

    out = dict(center=ratios(lo_c, hi_c))
    for d, name in enumerate(("Vx", "Vy", "Vz")[:nd]):
        h = 0.5 * (xv[d][1] - xv[d][0])
        lo, hi = list(lo_c), list(hi_c)
        lo[d] = np.maximum(xv[d] - h, xv[d][0])
        hi[d] = np.minimum(xv[d] + h, xv[d][-1])
        out[name] = ratios(lo, hi)
    return out


MULTIPHASE_RHEOLOGY = (
    dict(k=3.0, Cp=1.2e3, Hr=1.0e-6, density=dict(kind="PT", rho0=3.0e3, alpha=1.5e-5, beta=0.0, T0=0.0, P0=0.0)),
    dict(k=3.0, Cp=1.2e3, Hr=1.0e-7, density=dict(kind="PT", rho0=3.3e3, alpha=1.5e-5, beta=0.0, T0=0.0, P0=0.0)),
)


def diffusion2d_multiphase(n=32, *, lx=100.0e3, ly=100.0e3, iterMax=1000, nout=10, sharp=False) -> Setup:
    """diffusion_2D of test/test_diffusion2D_multiphase.jl:82-185: the geotherm / +100 K disc of test_diffusion2D.jl with two phases (disc = phase 2)
    that differ in ρ0 and radioactive heat; H = 0; ϵ = 1e-5, CFL = 0.95/√2; 20 steps of 50 kyr with iterMax = 1e3, nout = 10.  Phase ratios:
    area fractions of the disc (sharp=True: 0/1 by the cell-centre test) in place of the reference's randomly seeded particles."""
    nx, ny = (n, n) if isinstance(n, int) else tuple(n)
    kyr = 1.0e3 * 3600 * 24 * 365.25
    dt = 50 * kyr
    init_global_grid(nx, ny, 1)
    ni, li = (nx, ny), (lx, ly)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = Geometry(ni, li, origin=(0.0, -ly))
    arr = {k: fzeros_np(s) for k, s in thermal_shapes2d(nx, ny).items() if k not in ("K", "rhoCp")}
    z = grid.xci[1]
    arr["T"][:, 1:-1] = (z * (1900.0 - 1600.0) / z.min() + 1600.0)[None, :]
    bc = TemperatureBoundaryConditions(no_flux=dict(left=True, right=True, top=False, bot=False),
                                       constant_value=dict(left=True, right=True, top=300.0, bot=3500.0))
    centre, r = (lx / 2, -ly / 2), 10.0e3
    ph = ball_phase_ratios(grid, centre, r, s=1 if sharp else 8)
    arr["P"] = fzeros_np(ni)
    CFL = 0.95 / math.sqrt(2)
    return Setup(ni=ni, arrays=arr, grid=grid, pt=dict(eps=1.0e-5, CFL=CFL, max_lxyz=max(li), Vpdtau=min(di) * CFL), dt=dt, flow_bcs=bc,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False),
                 extra=dict(li=li, di=di, rheology=MULTIPHASE_RHEOLOGY, phase_ratios=ph, nt=int(math.ceil(1.0e3 * kyr / dt)),
                            perturbation=dict(δT=100.0, r=r, xc=centre[0], yc=centre[1])))
