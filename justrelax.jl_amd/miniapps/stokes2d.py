"""2D Stokes benchmark inputs (visco-elastic driver, Stokes2D.jl:181-325).

solcx2d           -- miniapps/benchmarks/stokes2D/solcx/SolCx.jl:54-140
solkz2d           -- miniapps/benchmarks/stokes2D/solkz/SolKz.jl:47-125
elastic_buildup2d -- miniapps/benchmarks/stokes2D/elastic_buildup/Elastic_BuildUp.jl:16-116
random_fields2d   -- kernel-parity inputs with finite dt, G, K.
"""
from __future__ import annotations

import math

import numpy as np

from ..arrays import PTStokesCoeffs, VelocityBoundaryConditions
from ..grid import Geometry, init_global_grid, nx_g, ny_g
from .common import Setup, alloc_stokes

_F4 = ("left", "right", "top", "bot")


def _smooth2d(A2, A, fact=1.0):
    """@inn(A2) = @inn(A) + 1.0/4.1/fact*(@d2_xi(A)+@d2_yi(A))   (SolCx.jl:6-11)"""
    c = A[1:-1, 1:-1]
    d2x = (A[2:, 1:-1] - c) - (c - A[:-2, 1:-1])
    d2y = (A[1:-1, 2:] - c) - (c - A[1:-1, :-2])
    A2[1:-1, 1:-1] = c + 1.0 / 4.1 / fact * (d2x + d2y)


def _free_slip2d_host(arr):
    Vx, Vy = arr["Vx"], arr["Vy"]
    Vx[:, 0] = Vx[:, 1]; Vx[:, -1] = Vx[:, -2]
    Vy[0, :] = Vy[1, :]; Vy[-1, :] = Vy[-2, :]


def solcx2d(n=32, *, Δη=1.0e6, iterMax=500_000, nout=5000) -> Setup:
    ni = (n, n)
    init_global_grid(n, n, 1)
    li = (1.0, 1.0)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0))
    arr = alloc_stokes(ni)
    pt = PTStokesCoeffs(li, di, CFL=1 / math.sqrt(2.1), ϵ_abs=1.0e-8, ϵ_rel=1.0e-9)
    xc, yc = grid.xci
    X, Y = np.meshgrid(xc, yc, indexing="ij")
    eta_stokes = np.asfortranarray(np.where(X <= 0.5, 1.0, Δη))       # solCx_viscosity :13-33
    arr["fy"][...] = -np.sin(math.pi * Y) * np.cos(math.pi * X) * 1    # solCx_density :35-52, g = 1
    arr["G"][...] = np.inf
    arr["K"][...] = np.inf
    # 5 ping-pong smoothing passes; the array the solver sees (stokes.viscosity.η) ends up with the
    # 4x-smoothed field (SolCx.jl:88-109)
    a, b = eta_stokes, eta_stokes.copy(order="F")       # a is "η" (aliases stokes.viscosity.η), b is η2
    for _ in range(5):
        _smooth2d(b, a, 1.0)
        b[0, :] = b[1, :]; b[-1, :] = b[-2, :]; b[:, 0] = b[:, 1]; b[:, -1] = b[:, -2]
        a, b = b, a
    arr["eta"][...] = eta_stokes
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F4})
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=0.1, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di))


def solkz2d(n=32, *, Δη=1.0e6, iterMax=150_000, nout=1000) -> Setup:
    ni = (n, n)
    init_global_grid(n, n, 1)
    li = (1.0, 1.0)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0))
    arr = alloc_stokes(ni)
    pt = PTStokesCoeffs(li, di, Re=5 * math.pi, CFL=1 / math.sqrt(2.1))
    xc, yc = grid.xci
    X, Y = np.meshgrid(xc, yc, indexing="ij")
    arr["eta"][...] = np.exp(math.log(Δη) * Y)
    arr["fy"][...] = -np.sin(2 * Y) * np.cos(3 * math.pi * X) * 1
    arr["G"][...] = np.inf
    arr["K"][...] = np.inf
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F4})
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=0.1, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di))


def elastic_buildup2d(n=32, *, lx=100.0e3, ly=100.0e3, η0=1.0e21, εbg=1.0e-14, G=10.0e9,
                      iterMax=150_000, nout=1000) -> Setup:
    """One Setup; the caller advances time with dt = 0.05 kyr (t < 10 kyr) else 1 kyr."""
    ni = (n, n)
    li = (lx, ly)
    di = tuple(l / m for l, m in zip(li, ni))
    init_global_grid(n, n, 1)
    grid = Geometry(ni, li, origin=(0.0, 0.0))
    arr = alloc_stokes(ni)
    pt = PTStokesCoeffs(li, di, ϵ_abs=1.0e-6, ϵ_rel=1.0e-6, CFL=1 / math.sqrt(2.1))
    arr["eta"][...] = η0
    arr["G"][...] = G
    arr["K"][...] = np.inf
    xc, yc = grid.xci
    xv, yv = grid.xvi
    arr["Vx"][:, 1:-1] = (εbg * xv)[:, None] * np.ones((1, len(yc)))       # pure_shear.jl:1-8
    arr["Vy"][1:-1, :] = (-εbg * yv)[None, :] * np.ones((len(xc), 1))
    _free_slip2d_host(arr)
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F4})
    kyr = 1.0e3 * 365.25 * 3600 * 24
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=0.05 * kyr, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False),
                 extra=dict(li=li, di=di, kyr=kyr, η0=η0, εbg=εbg, G=G))


def random_fields2d(ni, seed=20260821, *, dt=0.25, G=1.0, K=2.0, iterMax=20, nout=5, bcs="free_slip") -> Setup:
    ni = tuple(ni)
    init_global_grid(ni[0], ni[1], 1)
    rng = np.random.default_rng(seed)
    arr = alloc_stokes(ni)
    for k, a in arr.items():
        if k not in ("eta", "K", "G"):
            a[...] = rng.uniform(-1.0, 1.0, size=a.shape)
    arr["Q"][...] = rng.uniform(-0.1, 0.1, size=ni)
    arr["eta"][...] = 10.0 ** rng.uniform(-3.0, 0.0, size=ni)
    arr["G"][...] = G * (1.0 + 0.5 * rng.uniform(0.0, 1.0, size=ni))
    arr["K"][...] = K * (1.0 + 0.5 * rng.uniform(0.0, 1.0, size=ni))
    li = (1.0, 1.3)
    di = tuple(l / n for l, n in zip(li, ni))
    pt = PTStokesCoeffs(li, di)
    on, off = {f: True for f in _F4}, {f: False for f in _F4}
    b = {"free_slip": VelocityBoundaryConditions(free_slip=on, no_slip=off),
         "no_slip": VelocityBoundaryConditions(free_slip=off, no_slip=on),
         "periodic": VelocityBoundaryConditions(free_slip=off, no_slip=off, periodic=on),
         "none": VelocityBoundaryConditions(free_slip=off, no_slip=off)}[bcs]
    grid = Geometry(ni, li)
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=dt, flow_bcs=b,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di))


def shearband2d(n=32, *, iterMax=50_000, nout=100, xvi=None) -> Setup:
    """ShearBand2D -- test/test_shearband2D.jl:61-175 (BASELINE config 5): two phases (matrix G0 = 1, circular inclusion
    Gi = 0.5 of radius 0.1), LinearViscous eta = 1, Kb = 4, DruckerPrager_regularised(C = 1.6/cos30, phi = 30, psi = 0,
    eta_vp = 8e-3), pure shear eps_bg = 1, free slip, dt = eta0/G0/4.  Array names follow oracle.VEP_NAMES / jrx_vep2d_fields.
    xvi = (xv, yv): the same problem on the non-uniform grid of these vertices (miniapps/benchmarks/stokes2D/shear_band/ShearBand2D_refined.jl:43-52:
    grid = Geometry(xvi...), while di = li ./ ni still feeds PTStokesCoeffs, :50,110)."""
    nx = ny = n
    ni, li = (nx, ny), (1.0, 1.0)
    init_global_grid(nx, ny, 1)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = Geometry(ni, li, origin=(0.0, 0.0)) if xvi is None else Geometry.from_vertices(xvi)
    if xvi is not None and grid.ni != ni:
        raise ValueError("xvi must hold n + 1 vertices per dimension")
    τ_y, ϕ, η0, G0, εbg, η_reg = 1.6, 30.0, 1.0, 1.0, 1.0, 8.0e-3
    Gi = G0 / (6.0 - 4.0)
    dt = η0 / G0 / 4.0
    Cgp = τ_y / math.cos(math.radians(ϕ))
    phases = [dict(eta=η0, G=G0, Kb=4.0, C=Cgp, phi_deg=ϕ, psi_deg=0.0, eta_vp=η_reg),
              dict(eta=η0, G=Gi, Kb=4.0, C=Cgp, phi_deg=ϕ, psi_deg=0.0, eta_vp=η_reg)]
    c, v = (nx, ny), (nx + 1, ny + 1)
    shapes = {k: c for k in ("P", "P0", "divV", "Q", "exx", "eyy", "exy_c", "eplxx", "eplyy", "eplxy_c", "dexy_c", "txx", "tyy", "txy_c", "tII",
                             "toxx", "toyy", "toxy_c", "eta", "eta_vep", "EII_pl", "evol_pl", "EVol_pl", "fx", "fy", "RP")}
    shapes.update({k: v for k in ("exy", "eplxy", "dexy", "txy", "toxy", "eta_v", "omega_xy")})
    shapes.update(Vx=(nx + 1, ny + 2), Vy=(nx + 2, ny + 1), Ux=(nx + 1, ny + 2), Uy=(nx + 2, ny + 1), Rx=(nx - 1, ny), Ry=(nx, ny - 1),
                  phase_c=(2, nx, ny), phase_v=(2, nx + 1, ny + 1))
    arr = {k: np.zeros(s, dtype=np.float64, order="F") for k, s in shapes.items()}
    radius, ox, oy = 0.1, 0.5, 0.5
    for name, (xs, ys) in (("phase_c", grid.xci), ("phase_v", grid.xvi)):       # init_phases! :37-58
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        outside = ((X - ox) ** 2 + (Y - oy) ** 2) > radius ** 2
        arr[name][0] = np.where(outside, 1.0, 0.0)
        arr[name][1] = np.where(outside, 0.0, 1.0)
    arr["eta"][...] = η0                                                      # compute_viscosity!: linear viscous
    arr["eta_v"][...] = η0
    xv, yv = grid.xvi
    arr["Vx"][...] = (xv * εbg)[:, None] * np.ones((1, ny + 2))               # :145-146 (ghost rows included)
    arr["Vy"][...] = (-yv * εbg)[None, :] * np.ones((nx + 2, 1))
    _free_slip2d_host(arr)
    pt = PTStokesCoeffs(li, di, ϵ_rel=1.0e-6, CFL=0.75 / math.sqrt(2.1))
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F4}, no_slip={f: False for f in _F4})
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=dt, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False, viscosity_cutoff=(-np.inf, np.inf)),
                 extra=dict(li=li, di=di, phases=phases, εbg=εbg, G0=G0, η0=η0))


def shearheating2d(n=32, *, iterMax=75_000, nout=1000) -> Setup:
    """Shearheating2D -- test/test_shearheating2D.jl:66-232 without the particles (phase ratios from 8 x 8 sample points per cell / vertex area): 70 x 40 km box,
    dislocation-creep matrix and inclusion of Duretz et al. 2014 (Shearheating_rheology.jl:6-7; no elastic or plastic element), disc of radius 3 km at 40 km depth,
    T = 673 K, lithostatic initial pressure (:60-63), compression at εbg = 5e-14 / s, free slip, solve! with dt = Inf, ϵ_abs = ϵ_rel = 1e-5,
    viscosity_cutoff = (-Inf, Inf).  args.T is the ghosted thermal.T (arrays["T"], ni .+ 2)."""
    nx = ny = n
    ni, li = (nx, ny), (70.0e3, 40.0e3)
    init_global_grid(nx, ny, 1)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = Geometry(ni, li, origin=(0.0, -li[1]))
    inf = float("inf")
    common = dict(G=inf, Kb=inf, density=dict(kind="constant", rho0=2700.0), conductivity=2.5, heat_capacity=1050.0, shear_heat=1.0)
    phases = [dict(common, g=9.81, creep=dict(kind="dislocation", A=3.2e-20, n=3.0, E=276.0e3, V=0.0, R=8.3145)),
              dict(common, creep=dict(kind="dislocation", A=3.16e-26, n=3.3, E=186.0e3, V=0.0, R=8.3145))]
    arr = {k: np.zeros(shp, dtype=np.float64, order="F") for k, shp in _vep_shapes2d(nx, ny, 2).items()}
    ox, depth0, radius = li[0] / 2, 40.0e3, 3.0e3
    sub = (np.arange(8) + 0.5) / 8 - 0.5
    for name, (xs, ys) in (("phase_c", grid.xci), ("phase_v", grid.xvi)):
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        frac = np.zeros(X.shape)
        for a in sub:
            for b in sub:
                frac += ((X + a * di[0] - ox) ** 2 + (-(Y + b * di[1]) - depth0) ** 2) <= radius ** 2
        frac /= 64.0
        arr[name][0] = 1.0 - frac
        arr[name][1] = frac
    arr["T"] = np.asfortranarray(np.full((nx + 2, ny + 2), 273.0 + 400.0))
    yc = grid.xci[1]
    arr["fy"][...] = 2700.0 * 9.81                                             # compute_ρg!(ρg[2], phase_ratios, rheology, args) :118
    arr["P"][...] = np.abs(arr["fy"] * yc[None, :]) * (yc < 0.0)[None, :]
    εbg = 5.0e-14
    xv, yv = grid.xvi
    arr["Vx"][...] = (-(xv - li[0] / 2) * εbg)[:, None] * np.ones((1, ny + 2))                 # :134-135
    arr["Vy"][...] = ((li[1] - np.abs(yv)) * εbg)[None, :] * np.ones((nx + 2, 1))
    _free_slip2d_host(arr)
    arr["eta"][...] = 1.0e20             # overwritten by compute_viscosity!(stokes, phase_ratios, args, rheology, (-Inf, Inf)) :122
    arr["eta_v"][...] = 1.0e20
    pt = PTStokesCoeffs(li, di, ϵ_abs=1.0e-5, ϵ_rel=1.0e-5, CFL=0.9 / math.sqrt(2.1))
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F4}, no_slip={f: False for f in _F4})
    κ = 4.0 / (1050.0 * 2700.0)
    dt_diff = 0.5 * min(di) ** 2 / κ / 2.01
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=inf, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False, viscosity_cutoff=(-np.inf, np.inf)),           # :172
                 extra=dict(li=li, di=di, phases=phases, εbg=εbg, dt_diff=dt_diff))


def sinking_block2d(n=32, *, iterMax=150_000, nout=1000, sub=16) -> Setup:
    """Sinking_Block2D -- test/test_sinking_block.jl:93-203: a 500 km box, a 100 km square block (LinearViscous 1e23, ρ = 3300) at 400 km height in a mantle
    (1e21, ρ = 3200), g = 9.81, no elasticity, free slip, lithostatic initial pressure, dt = 1; the multiphase visco-elasto-plastic 2D solve! with a purely
    viscous table.  The reference seeds particles to get the phase ratios; here they are the area fractions of the block in each cell / around each vertex
    (sub x sub sampling)."""
    nx = ny = n
    ly = 500.0e3
    ni, li = (nx, ny), (ly, ly)
    init_global_grid(nx, ny, 1)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = Geometry(ni, li, origin=(0.0, -ly))
    phases = [dict(eta=1.0e21, G=float("inf"), Kb=float("inf"), g=9.81, density=dict(kind="constant", rho0=3.2e3)),
              dict(eta=1.0e23, G=float("inf"), Kb=float("inf"), density=dict(kind="constant", rho0=3.3e3))]
    arr = {k: np.zeros(shp, dtype=np.float64, order="F") for k, shp in _vep_shapes2d(nx, ny, 2).items()}
    xc0, depth0, r = 250.0e3, 100.0e3, 50.0e3        # (x - xc)^2 <= r^2 and (depth - yc)^2 <= r^2 with yc = |-(ly - 400 km)| (:139-146, :66)

    def frac(lo_x, hi_x, lo_y, hi_y):
        t = (np.arange(sub) + 0.5) / sub
        X = lo_x[:, None] + (hi_x - lo_x)[:, None] * t[None, :]                 # (n, sub)
        Y = lo_y[:, None] + (hi_y - lo_y)[:, None] * t[None, :]
        inx = ((X - xc0) ** 2 <= r ** 2).mean(axis=1)
        iny = ((-Y - depth0) ** 2 <= r ** 2).mean(axis=1)
        return inx[:, None] * iny[None, :]
    xv, yv = grid.xvi
    fc = frac(xv[:-1], xv[1:], yv[:-1], yv[1:])
    hx, hy = 0.5 * di[0], 0.5 * di[1]
    fv = frac(np.maximum(xv - hx, xv[0]), np.minimum(xv + hx, xv[-1]), np.maximum(yv - hy, yv[0]), np.minimum(yv + hy, yv[-1]))
    arr["phase_c"][0], arr["phase_c"][1] = 1.0 - fc, fc
    arr["phase_v"][0], arr["phase_v"][1] = 1.0 - fv, fv
    # compute_ρg!(ρg[2], phase_ratios, rheology, args) (:155), init_P! (:86-89,156), compute_viscosity! (:162: harmonic mean of the phase viscosities)
    arr["fy"][...] = (3.2e3 * arr["phase_c"][0] + 3.3e3 * arr["phase_c"][1]) * 9.81
    arr["P"][...] = arr["fy"] * np.abs(grid.xci[1])[None, :]
    def visc(r2):
        e = 1.0 / (r2[0] / 1.0e21 + r2[1] / 1.0e23)
        e[r2[0] > 0.999] = 1.0e21
        e[r2[1] > 0.999] = 1.0e23
        return e
    arr["eta"][...] = visc(arr["phase_c"])
    arr["eta_v"][...] = visc(arr["phase_v"])
    pt = PTStokesCoeffs(li, di, ϵ_rel=1.0e-5, CFL=0.95 / math.sqrt(2.1))
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F4}, no_slip={f: False for f in _F4})
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=1.0, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False, viscosity_cutoff=(-np.inf, np.inf)),
                 extra=dict(li=li, di=di, phases=phases))


def _thermal_bcs_host(T, bc):
    """thermal_bcs!(T, bc) on a host array (BoundaryConditions.jl:39-53): constant value (constant_value.jl:1-13; `2*true - T` for a
    face given as `true`), then no-flux copies (free_slip.jl:72-84); 2D: bot <-> j = 1"""
    cv, nf = bc.constant_value, bc.no_flux
    on = lambda v: v is not False and v is not None
    if on(cv.get("bot")): T[:, 0] = 2 * float(cv["bot"]) - T[:, 1]
    if on(cv.get("top")): T[:, -1] = 2 * float(cv["top"]) - T[:, -2]
    if on(cv.get("left")): T[0, :] = 2 * float(cv["left"]) - T[1, :]
    if on(cv.get("right")): T[-1, :] = 2 * float(cv["right"]) - T[-2, :]
    if nf.get("bot"): T[:, 0] = T[:, 1]
    if nf.get("top"): T[:, -1] = T[:, -2]
    if nf.get("left"): T[0, :] = T[1, :]
    if nf.get("right"): T[-1, :] = T[-2, :]


def thermal_convection2d(n=32, *, ar=8, iterMax=150_000, nout=1000) -> Setup:
    """The Stokes problem of test/test_WENO5.jl:105-245 (thermal_convection2D, `thermal_perturbation = :circular`): a 2890 km deep box of
    aspect ratio `ar`, half-space-cooling temperature with a +10 % circular anomaly, a single MaterialParams with an Arrhenius
    CustomRheology (test_WENO5.jl:25-42, depth = 0) + elasticity (G = 70 GPa, ν = 0.5), PT_Density(ρ0 = 3100, α = 1.5e-5, β = 0),
    g = 9.81, lithostatic initial pressure, free slip.  This is the input of the single-phase solve! (Stokes2D.jl:345-557); the WENO
    advection and the heat-diffusion step of the script are not part of it.  arrays["T"] is thermal.T (ghosted)."""
    from ..arrays import TemperatureBoundaryConditions
    nx = ny = n
    ly = 2890.0e3
    lx = ly * ar
    ni, li = (nx, ny), (lx, ly)
    init_global_grid(nx, ny, 1)
    di = tuple(l / m for l, m in zip(li, ni))
    grid = Geometry(ni, li, origin=(0.0, -ly))
    xc, yc = grid.xci
    v_args = dict(η0=5.0e20, Ea=200.0e3, Va=2.6e-6, T0=1.6e3, R=8.3145, cutoff=(1.0e16, 1.0e25))
    G0 = 70.0e9
    phase = dict(eta=v_args["η0"], G=G0, Kb=float("inf"), g=9.81,
                 density=dict(kind="PT", rho0=3.1e3, alpha=1.5e-5, beta=0.0, T0=0.0),
                 creep=dict(kind="arrhenius", Ea=v_args["Ea"], Va=v_args["Va"], T0=v_args["T0"], R=v_args["R"], cutoff=v_args["cutoff"]))
    κ = 3.0 / (1.2e3 * 3.1e3)
    dt = 0.5 * min(di) ** 2 / κ / 2.01
    # temperature: init_T! (:60-70), thermal_bcs!, circular_perturbation! (:72-84)
    adiabat, Tp = 0.3, 1900.0
    Tm = Tp + adiabat * 2890
    Tmin, Tmax = 300.0, 3.5e3
    T = np.zeros((nx + 2, ny + 2), order="F")
    yr = 3600 * 24 * 365.25
    z = np.abs(yc)
    Ti = Tp + (Tm - Tp) / 2890.0e3 * z
    Ths = Tmin + (Tm - Tmin) * np.array([math.erf(v * 0.5 / (κ * (100.0e6 * yr)) ** 0.5) for v in z])
    T[1:-1, 1:-1] = np.minimum(Ti, Ths)[None, :]
    tbc = TemperatureBoundaryConditions(no_flux=dict(left=True, right=True, top=False, bot=False),
                                        constant_value=dict(left=True, right=True, top=Tmin, bot=Tmax))
    _thermal_bcs_host(T, tbc)
    X, Y = np.meshgrid(xc, yc, indexing="ij")
    T[1:-1, 1:-1][((X - 0.5 * lx) ** 2 + (Y + 0.75 * ly) ** 2) <= 150.0e3 ** 2] *= 10.0 / 100 + 1
    arr = {k: np.zeros(s, dtype=np.float64, order="F") for k, s in _vep_shapes2d(nx, ny, 1).items()}
    arr["T"] = T
    # compute_ρg!(ρg[2], rheology, args) reads args.T at [i, j] of the ghosted array (no shift), then init_P! (:52-55)
    rho = 3.1e3 * (1.0 - 1.5e-5 * (T[:nx, :ny] - 0.0) + 0.0 * (0.0 - 0.0))
    arr["fy"][...] = rho * 9.81
    arr["P"][...] = arr["fy"] * np.abs(yc)[None, :]
    # compute_viscosity!(stokes, args, rheology, (1e16, 1e24)): T at [i+1, j+1], P at [i, j]
    cutoff = (1.0e16, 1.0e24)
    Tc, P = T[1:-1, 1:-1], arr["P"]
    η = v_args["η0"] * np.exp((v_args["Ea"] + P * v_args["Va"]) / (v_args["R"] * Tc) - v_args["Ea"] / (v_args["R"] * v_args["T0"]))
    arr["eta"][...] = np.clip(np.clip(η, *v_args["cutoff"]), *cutoff)
    arr["phase_c"][...] = 1.0
    arr["phase_v"][...] = 1.0
    pt = PTStokesCoeffs(li, di, ϵ_rel=1.0e-4, CFL=0.8 / math.sqrt(2.1))
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F4}, no_slip={f: False for f in _F4})
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=dt, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False, viscosity_cutoff=cutoff),
                 extra=dict(li=li, di=di, rheology=phase, thermal_bc=tbc))


def _vep_shapes2d(nx, ny, nphase):
    c, v = (nx, ny), (nx + 1, ny + 1)
    shapes = {k: c for k in ("P", "P0", "divV", "Q", "exx", "eyy", "exy_c", "eplxx", "eplyy", "eplxy_c", "dexy_c", "txx", "tyy", "txy_c", "tII",
                             "toxx", "toyy", "toxy_c", "eta", "eta_vep", "EII_pl", "evol_pl", "EVol_pl", "fx", "fy", "RP")}
    shapes.update({k: v for k in ("exy", "eplxy", "dexy", "txy", "toxy", "eta_v", "omega_xy")})
    shapes.update(Vx=(nx + 1, ny + 2), Vy=(nx + 2, ny + 1), Ux=(nx + 1, ny + 2), Uy=(nx + 2, ny + 1), Rx=(nx - 1, ny), Ry=(nx, ny - 1),
                  phase_c=(nphase, nx, ny), phase_v=(nphase, nx + 1, ny + 1))
    return shapes
