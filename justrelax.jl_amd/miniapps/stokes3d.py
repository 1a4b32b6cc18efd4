"""3D Stokes benchmark inputs.

solvi3d        -- miniapps/benchmarks/stokes3D/solvi/SolVi3D.jl:14-102 (spherical weak inclusion, pure shear)
taylor_green3d -- miniapps/benchmarks/stokes3D/taylor_green/TaylorGreen.jl:11-136 (analytic body force)
random_fields3d -- kernel-parity inputs with finite dt, G, K so that every elastic / compressible
                   term is exercised (SolVi/Taylor-Green have dt = Inf).
"""
from __future__ import annotations

import math

import numpy as np

from ..arrays import PTStokesCoeffs, VelocityBoundaryConditions
from ..grid import Geometry, global_grid, init_global_grid, nx_g, ny_g, nz_g, OVERLAP
from .common import Setup, alloc_stokes

_F6 = ("left", "right", "front", "back", "top", "bot")


def _smooth3d(A2, A, fact=1.0):
    """@inn(A2) = @inn(A) + 1/6.1/fact*(@d2_xi(A)+@d2_yi(A)+@d2_zi(A))  (SolVi3D.jl:9-12;
    ParallelStencil.FiniteDifferences3D: d2_xi = (A[i+1]-A[i]) - (A[i]-A[i-1]) on inner points)."""
    c = A[1:-1, 1:-1, 1:-1]
    d2x = (A[2:, 1:-1, 1:-1] - c) - (c - A[:-2, 1:-1, 1:-1])
    d2y = (A[1:-1, 2:, 1:-1] - c) - (c - A[1:-1, :-2, 1:-1])
    d2z = (A[1:-1, 1:-1, 2:] - c) - (c - A[1:-1, 1:-1, :-2])
    A2[1:-1, 1:-1, 1:-1] = c + 1.0 / 6.1 / fact * (d2x + d2y + d2z)


def solvi_viscosity(ni, di, li, rc, η0, ηi, update_halo=None):
    """SolVi3D.jl:14-43.  The reference places the inclusion with *local* indices; here the rank
    offset coord*(n-2) is added so that a decomposed run sees one global inclusion (identical on
    one rank)."""
    gg = global_grid()
    off = [gg.coords[d] * (ni[d] - OVERLAP) if gg.initialized else 0 for d in range(3)]
    ix = [(np.arange(ni[d]) + off[d]) * di[d] + 0.5 * di[d] - 0.5 * li[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ix, indexing="ij")
    η = np.full(ni, η0, dtype=np.float64, order="F")
    η[np.sqrt(X ** 2 + Y ** 2 + Z ** 2) <= rc] = ηi
    η2 = η.copy(order="F")
    for _ in range(10):
        _smooth3d(η2, η, 1.0)
        η, η2 = η2, η
        if update_halo is not None:
            update_halo(η)
    return η


def pureshear_bc3d(arr, xci, xvi, εbg):
    """src/boundaryconditions/pure_shear.jl:10-30 (note: Vy takes its y-coordinate from xv and
    Vz its y from xc -- reference quirk, exact for cubic domains)."""
    xv, yv, zv = xvi
    xc, yc, zc = xci
    arr["Vx"][:, 1:-1, 1:-1] = (εbg * xv)[:, None, None] * np.ones((1, len(yc), len(zc)))
    arr["Vy"][1:-1, :, 1:-1] = (εbg * xv)[None, :, None] * np.ones((len(xc), 1, len(zc)))
    arr["Vz"][1:-1, 1:-1, :] = (-εbg * zv)[None, None, :] * np.ones((len(xc), len(xc), 1))


def _free_slip3d_host(arr):
    """flow_bcs!(free_slip all faces) on host arrays -- face groups in the reference's source
    order (free_slip.jl:15-70): front/back, top/bot (k=1/k=end), left/right."""
    Vx, Vy, Vz = arr["Vx"], arr["Vy"], arr["Vz"]
    Vx[:, 0, :] = Vx[:, 1, :]; Vz[:, 0, :] = Vz[:, 1, :]
    Vx[:, -1, :] = Vx[:, -2, :]; Vz[:, -1, :] = Vz[:, -2, :]
    Vx[:, :, 0] = Vx[:, :, 1]; Vy[:, :, 0] = Vy[:, :, 1]
    Vx[:, :, -1] = Vx[:, :, -2]; Vy[:, :, -1] = Vy[:, :, -2]
    Vy[0, :, :] = Vy[1, :, :]; Vz[0, :, :] = Vz[1, :, :]
    Vy[-1, :, :] = Vy[-2, :, :]; Vz[-1, :, :] = Vz[-2, :, :]


def solvi3d(n=16, *, Δη=1.0e-3, lx=10.0, ly=10.0, lz=10.0, rc=1.0, εbg=1.0, iterMax=5000, nout=100,
            init_grid=True) -> Setup:
    """solVi3D(; nx, ny, nz, ...) -- SolVi3D.jl:45-129.  `n` is the *local* cell count per dim."""
    ni = (n, n, n) if isinstance(n, int) else tuple(n)
    if init_grid:
        init_global_grid(*ni)
    li = (lx, ly, lz)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g(), nz_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0, 0.0))
    arr = alloc_stokes(ni)
    pt = PTStokesCoeffs(li, di, CFL=1 / math.sqrt(3))
    arr["eta"][...] = solvi_viscosity(ni, di, li, rc, 1.0, Δη)
    arr["G"][...] = 1.0
    arr["K"][...] = np.inf
    pureshear_bc3d(arr, grid.xci, grid.xvi, εbg)
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F6}, no_slip={f: False for f in _F6})
    gg = global_grid()
    if gg.nprocs == 1:
        _free_slip3d_host(arr)
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=float("inf"), flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di))


def taylor_green3d(n=16, *, iterMax=100_000, nout=1000, init_grid=True) -> Setup:
    """taylorGreen(; nx, ny, nz) -- TaylorGreen.jl:80-152."""
    ni = (n, n, n)
    if init_grid:
        init_global_grid(*ni)
    li = (1.0, 1.0, 1.0)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g(), nz_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0, 0.0))
    arr = alloc_stokes(ni)
    pt = PTStokesCoeffs(li, di, CFL=1 / math.sqrt(3))
    arr["eta"][...] = 1.0
    xc, yc, zc = grid.xci
    X, Y, Z = np.meshgrid(xc, yc, zc, indexing="ij")
    arr["fx"][...] = 36 * math.pi ** 2 * np.cos(2 * math.pi * X) * np.sin(2 * math.pi * Y) * np.sin(2 * math.pi * Z)
    arr["G"][...] = np.inf
    arr["K"][...] = np.inf
    # velocity!(stokes, xci, xvi): analytic velocity on every outer plane, zero inside (TaylorGreen.jl:25-78)
    xv, yv, zv = grid.xvi
    d = [c[1] - c[0] for c in grid.xci]
    xce, yce, zce = [np.linspace(c[0] - dd, c[-1] + dd, len(c) + 2) for c, dd in zip(grid.xci, d)]
    vx = lambda x, y, z: -2 * np.cos(2 * math.pi * x) * np.sin(2 * math.pi * y) * np.sin(2 * math.pi * z)
    vy = lambda x, y, z: np.sin(2 * math.pi * x) * np.cos(2 * math.pi * y) * np.sin(2 * math.pi * z)
    vz = lambda x, y, z: np.sin(2 * math.pi * x) * np.sin(2 * math.pi * y) * np.cos(2 * math.pi * z)
    for name, fn, cs in (("Vx", vx, (xv, yce, zce)), ("Vy", vy, (xce, yv, zce)), ("Vz", vz, (xce, yce, zv))):
        A = arr[name]
        full = fn(*np.meshgrid(*cs, indexing="ij"))
        mask = np.zeros(A.shape, dtype=bool)
        mask[0], mask[-1], mask[:, 0], mask[:, -1], mask[:, :, 0], mask[:, :, -1] = (True,) * 6
        A[...] = np.where(mask, full, 0.0)
    none = {f: False for f in _F6}
    bcs = VelocityBoundaryConditions(free_slip=dict(none), no_slip=dict(none))
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=float("inf"), flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di))


def taylor_green_error_norms(arr, grid):
    """error_norms -- vizTaylorGreen.jl:100-119 : sqrt(Σ e² ΔV), pressures with mean removed."""
    xc, yc, zc = grid.xci
    xv, yv, zv = grid.xvi
    dV = float(np.prod(grid.di["center"]))
    s, c = lambda a: np.sin(2 * math.pi * a), lambda a: np.cos(2 * math.pi * a)
    m = lambda *cs: np.meshgrid(*cs, indexing="ij")
    X, Y, Z = m(xv, yc, zc); vx = -2 * c(X) * s(Y) * s(Z)
    X, Y, Z = m(xc, yv, zc); vy = s(X) * c(Y) * s(Z)
    X, Y, Z = m(xc, yc, zv); vz = s(X) * s(Y) * c(Z)
    X, Y, Z = m(xc, yc, zc); p = -6 * math.pi * s(X) * s(Y) * s(Z)
    L2 = lambda e: math.sqrt(float(np.sum(e * e)) * dV)
    P = arr["P"]
    return (L2((P - P.mean()) - (p - p.mean())), L2(arr["Vx"][:, 1:-1, 1:-1] - vx),
            L2(arr["Vy"][1:-1, :, 1:-1] - vy), L2(arr["Vz"][1:-1, 1:-1, :] - vz))


def random_fields3d(ni, seed=20260821, *, dt=0.25, G=1.0, K=2.0, iterMax=20, nout=5,
                    bcs="free_slip") -> Setup:
    """Kernel-parity inputs: fields ~U(-1,1), η = 10^U(-3,0), G, K finite, finite dt,
    Q ~ U(-0.1,0.1).  Every array the kernels read is non-trivial."""
    ni = tuple(ni)
    init_global_grid(*ni)
    rng = np.random.default_rng(seed)
    arr = alloc_stokes(ni)
    for k, a in arr.items():
        if k in ("eta", "K", "G"):
            continue
        a[...] = rng.uniform(-1.0, 1.0, size=a.shape)
    arr["Q"][...] = rng.uniform(-0.1, 0.1, size=ni)
    arr["eta"][...] = 10.0 ** rng.uniform(-3.0, 0.0, size=ni)
    arr["G"][...] = G * (1.0 + 0.5 * rng.uniform(0.0, 1.0, size=ni))
    arr["K"][...] = K * (1.0 + 0.5 * rng.uniform(0.0, 1.0, size=ni))
    li = (1.0, 1.3, 0.9)
    di = tuple(l / n for l, n in zip(li, ni))
    pt = PTStokesCoeffs(li, di)
    on = {f: True for f in _F6}
    off = {f: False for f in _F6}
    if bcs == "free_slip":
        b = VelocityBoundaryConditions(free_slip=on, no_slip=off)
    elif bcs == "no_slip":
        b = VelocityBoundaryConditions(free_slip=off, no_slip=on)
    elif bcs == "periodic":
        b = VelocityBoundaryConditions(free_slip=off, no_slip=off, periodic=on)
    elif bcs == "slip_mix":      # free slip and no slip on different faces, one face left alone (prescribed values), nothing periodic
        b = VelocityBoundaryConditions(free_slip=dict(off, left=True, right=True, front=True), no_slip=dict(off, top=True, bot=True))
    elif bcs == "mixed":
        b = VelocityBoundaryConditions(free_slip=dict(off, left=True, right=True), no_slip=dict(off, top=True, bot=True),
                                       periodic=dict(off, front=True, back=True))
    else:
        b = VelocityBoundaryConditions(free_slip=off, no_slip=off)
    grid = Geometry(ni, li)
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=dt, flow_bcs=b,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di))


def plane_strain3d(s2: Setup, nz=3, axis=2) -> Setup:
    """The 3D restatement of a 2D visco-elastic setup: every field uniform along the 3D axis `axis` (nz cells of the in-plane spacing there), the
    velocity along it 0, the out-of-plane stresses zero, free slip on its two faces, and the 2D run's own PT coefficients.  The 2D x and y take the two
    remaining 3D axes in order (axis = 2: x, y -- the fields are uniform along z; axis = 1: x, z; axis = 0: y, z), and the 2D components are renamed with
    them (axis = 1: τxy -> τxz, τyy -> τzz, Vy -> Vz, fy -> fz; axis = 0: τxx -> τyy, τyy -> τzz, τxy -> τyz, Vx -> Vy, Vy -> Vz).
    With ∂/∂(axis) = 0 the 3D iteration (Stokes3D.jl:78-121) is algebraically the 2D one (Stokes2D.jl:229-275) -- same ∇V, the same ε_xx = ∂Vx/∂x − ∇V/3,
    the same clamped shear-node averages, ητ of a uniform-along-axis η -- so a 3D run must reproduce the pinned 2D run to round-off (the kernels differ in fma
    placement and summation order; the terms that vanish are exact zeros).  This is how the elastic (τ_o, 1/(G dt)) and compressible (1/(K dt)) terms of the
    3D kernels get a numeric anchor: every 3D test of the reference runs with dt = Inf and/or G = K = Inf, the 2D elastic build-up
    (miniapps/benchmarks/stokes2D/elastic_buildup/Elastic_BuildUp.jl:4,55-56,75-86, test/test_stokes_elastic_buildup.jl:47-54) does not.  The three
    orientations together anchor all six stress components -- axis = 0 and 1 put the 2D shear stress on τyz / τxz and a 2D normal stress on τzz, the
    components the z-marching kernels carry from plane to plane."""
    if axis not in (0, 1, 2):
        raise ValueError("axis must be 0, 1 or 2")
    u = axis
    a, b = [d for d in range(3) if d != u]            # the 3D axes of the 2D x and y
    n2 = s2.ni
    ni = [0, 0, 0]
    ni[a], ni[b], ni[u] = n2[0], n2[1], nz
    ni = tuple(ni)
    du = s2.extra["di"][0]
    li, di, org = [0.0] * 3, [0.0] * 3, [0.0] * 3
    li[a], li[b], li[u] = s2.extra["li"][0], s2.extra["li"][1], du * nz
    di[a], di[b], di[u] = s2.extra["di"][0], s2.extra["di"][1], du
    org[a], org[b] = s2.grid.origin[0], s2.grid.origin[1]
    init_global_grid(*ni)
    grid = Geometry(ni, tuple(li), origin=tuple(org))
    arr = alloc_stokes(ni)
    a2 = s2.arrays
    c = "xyz"
    nrm = lambda d: c[d] + c[d]
    shear = {(1, 2): "yz", (0, 2): "xz", (0, 1): "xy"}[(a, b)]
    put = lambda k3, k2: arr[k3].__setitem__(Ellipsis, np.expand_dims(a2[k2], u))     # broadcast along the uniform axis (ghost planes of V included)
    for k in ("P", "P0", "Q", "eta", "K", "G"):
        put(k, k)
    for pre in ("t", "to"):
        put(pre + nrm(a), pre + "xx")
        put(pre + nrm(b), pre + "yy")
        put(pre + shear, pre + "xy")
    put("V" + c[a], "Vx"); put("V" + c[b], "Vy")
    put("f" + c[a], "fx"); put("f" + c[b], "fy")
    # faces: 2D left / right are x-lo / x-hi, bot / top are y-lo (j = 1) / y-hi; in 3D the x faces are left / right, the y faces front / back, and the z faces
    # carry the reference's inconsistent names: free_slip `top` is k = 1 and `bot` k = end, no_slip `bot` is k = 1 and `top` k = end
    lohi = {"fs": ({0: "left", 1: "front", 2: "top"}, {0: "right", 1: "back", 2: "bot"}),
            "ns": ({0: "left", 1: "front", 2: "bot"}, {0: "right", 1: "back", 2: "top"})}
    b2 = s2.flow_bcs
    fs = {f: False for f in _F6}
    ns = {f: False for f in _F6}
    for tag, tgt, src in (("fs", fs, b2.free_slip), ("ns", ns, b2.no_slip)):
        lo, hi = lohi[tag]
        tgt[lo[a]] = bool(src.get("left", False)); tgt[hi[a]] = bool(src.get("right", False))
        tgt[lo[b]] = bool(src.get("bot", False)); tgt[hi[b]] = bool(src.get("top", False))
    fs[lohi["fs"][0][u]] = fs[lohi["fs"][1][u]] = True                                  # the faces of the uniform axis slip freely
    bcs = VelocityBoundaryConditions(free_slip=fs, no_slip=ns)
    names = {"P": "P", "P0": "P0", "Q": "Q", "eta": "eta", "K": "K", "G": "G", "divV": "divV", "RP": "RP",
             "Vx": "V" + c[a], "Vy": "V" + c[b], "Rx": "R" + c[a], "Ry": "R" + c[b], "fx": "f" + c[a], "fy": "f" + c[b]}
    for pre in ("t", "to", "e"):
        names[pre + "xx"], names[pre + "yy"], names[pre + "xy"] = pre + nrm(a), pre + nrm(b), pre + shear
    return Setup(ni=ni, arrays=arr, grid=grid, pt=s2.pt, dt=s2.dt, flow_bcs=bcs, kwargs=dict(s2.kwargs),
                 extra=dict(s2.extra, li=tuple(li), di=tuple(di), axis=u, names=names, out_of_plane=dict(V="V" + c[u], tn=nrm(u), shear=[v for v in ("yz", "xz", "xy") if v != shear])))


def burstedde3d(n=16, *, β=10.0, iterMax=100_000, nout=1000) -> Setup:
    """Burstedde et al. (2013) manufactured solution -- miniapps/benchmarks/stokes3D/burstedde/Burstedde.jl:10-215
    (test/test_stokes_burstedde.jl): η = exp(1 − β Σ x(1−x)), analytical body forces, the analytical velocity prescribed on
    every face (no free-slip / no-slip face), net boundary flux removed; K = G = Inf, dt = Inf, CFL = 1/√3."""
    ni = (n, n, n)
    li = (1.0, 1.0, 1.0)
    init_global_grid(n, n, n)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g(), nz_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0, 0.0))
    arr = alloc_stokes(ni)
    xc, yc, zc = grid.xci
    X, Y, Z = np.meshgrid(xc, yc, zc, indexing="ij")
    η = np.exp(1 - β * (X * (1 - X) + Y * (1 - Y) + Z * (1 - Z)))               # _viscosity! :10-14
    arr["eta"][...] = η
    dηdx, dηdy, dηdz = -β * (1 - 2 * X) * η, -β * (1 - 2 * Y) * η, -β * (1 - 2 * Z) * η
    x, y, z = X, Y, Z                                                            # body_forces :22-44
    fx = ((y * z + 3 * x**2 * y**3 * z) - η * (2 + 6 * x * y)) - dηdx * (2 + 4 * x + 2 * y + 6 * x**2 * y) - dηdy * (x + x**3 + y + 2 * x * y**2) \
        - dηdz * (-3 * z - 10 * x * y * z)
    fy = ((x * z + 3 * x**3 * y**2 * z) - η * (2 + 2 * x**2 + 2 * y**2)) - dηdx * (x + x**3 + y + 2 * x * y**2) \
        - dηdy * (2 + 2 * x + 4 * y + 4 * x**2 * y) - dηdz * (-3 * z - 5 * x**2 * z)
    fz = ((x * y + x**3 * y**3) - η * (-10 * y * z)) - dηdx * (-3 * z - 10 * x * y * z) - dηdy * (-3 * z - 5 * x**2 * z) \
        - dηdz * (-4 - 6 * x - 6 * y - 10 * x**2 * y)
    arr["fx"][...], arr["fy"][...], arr["fz"][...] = -fx, -fy, -fz
    arr["K"][...] = np.inf
    arr["G"][...] = np.inf
    # velocity! :46-104: analytical values on the outermost planes of every V array, zero inside
    xv, yv, zv = grid.xvi
    ext = lambda c, d: np.linspace(c[0] - d, c[-1] + d, len(c) + 2)
    dd = [c[1] - c[0] for c in grid.xci]
    xg, yg, zg = ext(xc, dd[0]), ext(yc, dd[1]), ext(zc, dd[2])
    vx = lambda x_, y_: x_ + x_**2 + x_ * y_ + x_**3 * y_
    vy = lambda x_, y_: y_ + x_ * y_ + y_**2 + x_**2 * y_**2
    vz = lambda x_, y_, z_: -2 * z_ - 3 * x_ * z_ - 3 * y_ * z_ - 5 * x_**2 * y_ * z_

    def shell(A, full):
        m = np.zeros(A.shape, dtype=bool)
        for d in range(3):
            idx = [slice(None)] * 3
            for e in (0, A.shape[d] - 1):
                idx[d] = e
                m[tuple(idx)] = True
        A[...] = np.where(m, full, 0.0)
    Xa, Ya, _ = np.meshgrid(xv, yg, zg, indexing="ij")
    shell(arr["Vx"], vx(Xa, Ya))
    Xa, Ya, _ = np.meshgrid(xg, yv, zg, indexing="ij")
    shell(arr["Vy"], vy(Xa, Ya))
    Xa, Ya, Za = np.meshgrid(xg, yg, zv, indexing="ij")
    shell(arr["Vz"], vz(Xa, Ya, Za))
    # remove_net_flux! :106-139
    Vx, Vy, Vz = arr["Vx"], arr["Vy"], arr["Vz"]
    Ax, Ay, Az = di[1] * di[2], di[0] * di[2], di[0] * di[1]
    flux = ((Vx[-1, 1:-1, 1:-1].sum() - Vx[0, 1:-1, 1:-1].sum()) * Ax + (Vy[1:-1, -1, 1:-1].sum() - Vy[1:-1, 0, 1:-1].sum()) * Ay
            + (Vz[1:-1, 1:-1, -1].sum() - Vz[1:-1, 1:-1, 0].sum()) * Az)
    δ = flux / (2 * (ni[1] * ni[2] * Ax + ni[0] * ni[2] * Ay + ni[0] * ni[1] * Az))
    Vx[0] += δ; Vx[-1] -= δ; Vy[:, 0] += δ; Vy[:, -1] -= δ; Vz[:, :, 0] += δ; Vz[:, :, -1] -= δ
    pt = PTStokesCoeffs(li, di, CFL=1 / math.sqrt(3))
    off = {f: False for f in _F6}
    bcs = VelocityBoundaryConditions(free_slip=off, no_slip=dict(off))
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=float("inf"), flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False), extra=dict(li=li, di=di, β=β))


def burstedde_error_norms(arr, grid, di):
    """error_norms (vizBurstedde.jl:96-117): (L2_p, L2_vx, L2_vy, L2_vz), each sqrt(Σ e² ΔV); pressures with their means removed"""
    (xc, yc, zc), (xv, yv, zv) = grid.xci, grid.xvi
    dV = float(np.prod(di))
    L2 = lambda e: math.sqrt(float((e * e).sum()) * dV)
    X, Y, Z = np.meshgrid(xv, yc, zc, indexing="ij")
    e_vx = arr["Vx"][:, 1:-1, 1:-1] - (X + X**2 + X * Y + X**3 * Y)
    X, Y, Z = np.meshgrid(xc, yv, zc, indexing="ij")
    e_vy = arr["Vy"][1:-1, :, 1:-1] - (Y + X * Y + Y**2 + X**2 * Y**2)
    X, Y, Z = np.meshgrid(xc, yc, zv, indexing="ij")
    e_vz = arr["Vz"][1:-1, 1:-1, :] - (-2 * Z - 3 * X * Z - 3 * Y * Z - 5 * X**2 * Y * Z)
    X, Y, Z = np.meshgrid(xc, yc, zc, indexing="ij")
    p = X * Y * Z + X**3 * Y**3 * Z - 5 / 32
    return L2((arr["P"] - arr["P"].mean()) - (p - p.mean())), L2(e_vx), L2(e_vy), L2(e_vz)


def vep_shapes3d(ni, nphase=2):
    """array name -> extent for the 3D VEP problem (names follow jrx_vep3d_fields / oracle.VEP3_NAMES)"""
    nx, ny, nz = ni
    c = tuple(ni)
    yz, xz, xy = (nx, ny + 1, nz + 1), (nx + 1, ny, nz + 1), (nx + 1, ny + 1, nz)
    names = ["P", "P0", "divV", "Q", "exx", "eyy", "ezz", "eyz_c", "exz_c", "exy_c", "eplxx", "eplyy", "eplzz", "eplyz_c", "eplxz_c", "eplxy_c",
             "deyz_c", "dexz_c", "dexy_c", "txx", "tyy", "tzz", "tyz_c", "txz_c", "txy_c", "tII", "toxx", "toyy", "tozz", "toyz_c", "toxz_c", "toxy_c",
             "eta", "eta_vep", "EII_pl", "evol_pl", "EVol_pl", "fx", "fy", "fz", "RP"]
    s = {k: c for k in names}
    for pre in ("e", "epl", "de", "t", "to", "omega_"):
        s[pre + "yz"], s[pre + "xz"], s[pre + "xy"] = yz, xz, xy
    s.update(Vx=(nx + 1, ny + 2, nz + 2), Vy=(nx + 2, ny + 1, nz + 2), Vz=(nx + 2, ny + 2, nz + 1),
             Ux=(nx + 1, ny + 2, nz + 2), Uy=(nx + 2, ny + 1, nz + 2), Uz=(nx + 2, ny + 2, nz + 1),
             Rx=(nx - 1, ny, nz), Ry=(nx, ny - 1, nz), Rz=(nx, ny, nz - 1),
             phase_c=(nphase,) + c, phase_yz=(nphase,) + yz, phase_xz=(nphase,) + xz, phase_xy=(nphase,) + xy)
    return s


def shearband3d(n=16, *, iterMax=150_000, nout=1000) -> Setup:
    """ShearBand3D -- test/test_shearband3D_MPI.jl:68-168: matrix (η0 = 1, G0 = 1) with a spherical inclusion (η0/10, Gi = 0.5) of
    radius 0.1, ν = 0.5 (Kb = Inf), DruckerPrager_regularised(C = 1.6/cos30, ϕ = 30, ψ = 0, η_vp = 1.25e-2), pure shear in x-z
    (εbg = 1), free slip, dt = η0/G0/4, PTStokesCoeffs(ϵ_rel = 1e-5, Re = 3, r = 0.7, CFL = 0.9/√3.1)."""
    ni = (n, n, n) if isinstance(n, int) else tuple(n)
    nx, ny, nz = ni
    li = (1.0, 1.0, 1.0)
    init_global_grid(nx, ny, nz)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g(), nz_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0, 0.0))
    τ_y, ϕ, η0, G0, εbg, η_reg = 1.6, 30.0, 1.0, 1.0, 1.0, 1.25e-2
    Gi = G0 / (6.0 - 4.0)
    dt = η0 / G0 / 4.0
    Cgp = τ_y / math.cos(math.radians(ϕ))
    inf = float("inf")
    phases = [dict(eta=η0, G=G0, Kb=inf, C=Cgp, phi_deg=ϕ, psi_deg=0.0, eta_vp=η_reg),
              dict(eta=η0 / 10, G=Gi, Kb=inf, C=Cgp, phi_deg=ϕ, psi_deg=0.0, eta_vp=η_reg)]
    arr = {k: np.zeros(s, dtype=np.float64, order="F") for k, s in vep_shapes3d(ni).items()}
    radius, o = 0.1, (0.5, 0.5, 0.5)
    (xc, yc, zc), (xv, yv, zv) = grid.xci, grid.xvi
    for name, (xs, ys, zs) in (("phase_c", (xc, yc, zc)), ("phase_yz", (xc, yv, zv)), ("phase_xz", (xv, yc, zv)), ("phase_xy", (xv, yv, zc))):
        X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")                       # init_phases! :41-65, evaluated at the node itself
        outside = ((X - o[0]) ** 2 + (Y - o[1]) ** 2 + (Z - o[2]) ** 2) > radius ** 2
        arr[name][0] = np.where(outside, 1.0, 0.0)
        arr[name][1] = np.where(outside, 0.0, 1.0)
    arr["eta"][...] = np.where(arr["phase_c"][0] > 0.5, η0, η0 / 10)           # compute_viscosity!: linear viscous
    arr["Vx"][...] = (xv * εbg)[:, None, None] * np.ones((1, ny + 2, nz + 2))   # :146-147
    arr["Vz"][...] = (-zv * εbg)[None, None, :] * np.ones((nx + 2, ny + 2, 1))
    _free_slip3d_host(arr)
    pt = PTStokesCoeffs(li, di, ϵ_rel=1.0e-5, Re=3.0, r=0.7, CFL=0.9 / math.sqrt(3.1))
    on = {f: True for f in _F6}
    bcs = VelocityBoundaryConditions(free_slip=on, no_slip={f: False for f in _F6})
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=dt, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False, viscosity_cutoff=(-np.inf, np.inf)),
                 extra=dict(li=li, di=di, phases=phases, εbg=εbg, G0=G0, η0=η0))


def shearheating3d(n=16, *, iterMax=100_000, nout=1000) -> Setup:
    """Shearheating3D -- test/test_shearheating3D.jl:62-162 without the particles (phase ratios from 4^3 sample points per node volume): 70 x 70 x 40 km box, dislocation-creep
    matrix and inclusion of Duretz et al. 2014 (miniapps/benchmarks/stokes3D/shear_heating/Shearheating_rheology.jl:6-7: no elastic, no plastic element), sphere of
    radius 3 km at 40 km depth, T = 673 K, lithostatic initial pressure (init_P! :55-58), compression at εbg = 5e-14 / s, free slip, solve! with dt = Inf,
    ϵ_rel = 1e-4, viscosity_cutoff = (1e18, 1e22).  args.T is the ghosted thermal.T (arrays["T"], ni .+ 2)."""
    ni = (n, n, n) if isinstance(n, int) else tuple(n)
    nx, ny, nz = ni
    li = (70.0e3, 70.0e3, 40.0e3)
    init_global_grid(nx, ny, nz)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g(), nz_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0, -li[2]))
    inf = float("inf")
    common = dict(G=inf, Kb=inf, density=dict(kind="constant", rho0=2700.0), conductivity=2.5, heat_capacity=1050.0, shear_heat=1.0)
    phases = [dict(common, g=9.81, creep=dict(kind="dislocation", A=3.2e-20, n=3.0, E=276.0e3, V=0.0, R=8.3145)),
              dict(common, creep=dict(kind="dislocation", A=3.16e-26, n=3.3, E=186.0e3, V=0.0, R=8.3145))]
    arr = {k: np.zeros(s, dtype=np.float64, order="F") for k, s in vep_shapes3d(ni).items()}
    o, radius = (li[0] / 2, li[1] / 2, 40.0e3), 3.0e3
    (xc, yc, zc), (xv, yv, zv) = grid.xci, grid.xvi
    for name, (xs, ys, zs) in (("phase_c", (xc, yc, zc)), ("phase_yz", (xc, yv, zv)), ("phase_xz", (xv, yc, zv)), ("phase_xy", (xv, yv, zc))):
        X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")                       # init_phases! (Shearheating_rheology.jl:58-80): depth = -z
        # the reference marks ~100 particles per cell and update_phase_ratios! counts them; here 4^3 sample points of the node's own cell-sized volume
        frac, sub = np.zeros(X.shape), (np.arange(4) + 0.5) / 4 - 0.5
        for a in sub:
            for b in sub:
                for c in sub:
                    frac += ((X + a * di[0] - o[0]) ** 2 + (Y + b * di[1] - o[1]) ** 2 + (-(Z + c * di[2]) - o[2]) ** 2) <= radius ** 2
        frac /= 64.0
        arr[name][0] = 1.0 - frac
        arr[name][1] = frac
    arr["T"] = np.asfortranarray(np.full((nx + 2, ny + 2, nz + 2), 273.0 + 400.0))
    arr["fz"][...] = 2700.0 * 9.81                                              # compute_ρg!(ρg[3], phase_ratios, rheology, args) :122
    arr["P"][...] = np.abs(arr["fz"] * zc[None, None, :]) * (zc < 0.0)[None, None, :]
    εbg = 5.0e-14
    arr["Vx"][...] = (-(xv - li[0] / 2) * εbg)[:, None, None] * np.ones((1, ny + 2, nz + 2))       # :142-144
    arr["Vy"][...] = (-(yv - li[1] / 2) * εbg)[None, :, None] * np.ones((nx + 2, 1, nz + 2))
    arr["Vz"][...] = ((li[2] - np.abs(zv)) * εbg)[None, None, :] * np.ones((nx + 2, ny + 2, 1))
    _free_slip3d_host(arr)
    arr["eta"][...] = 1.0e20             # overwritten by compute_viscosity!(stokes, phase_ratios, args, rheology, (-Inf, Inf)) :127
    pt = PTStokesCoeffs(li, di, ϵ_rel=1.0e-4, CFL=0.9 / math.sqrt(3.1))
    on = {f: True for f in _F6}
    bcs = VelocityBoundaryConditions(free_slip=on, no_slip={f: False for f in _F6})
    κ = 4.0 / (1050.0 * 2700.0)
    dt_diff = 0.5 * min(di) ** 3 / κ / 3.01                                     # :79 (as written)
    return Setup(ni=ni, arrays=arr, grid=grid, pt=pt, dt=inf, flow_bcs=bcs,
                 kwargs=dict(iterMax=iterMax, nout=nout, verbose=False, viscosity_cutoff=(1.0e18, 1.0e22)),
                 extra=dict(li=li, di=di, phases=phases, εbg=εbg, dt_diff=dt_diff))


def solvi3d_device(n, backend_tag, *, Δη=1.0e-3, li=(10.0, 10.0, 10.0), rc=1.0, εbg=1.0, update_halo=None):
    """SolVi3D built directly in device memory (for sizes whose host copy would not fit: 512³ needs
    ≈48 GB of fields).  Same construction as `solvi3d` (torch elementwise ops instead of numpy);
    returns (stokes, ρg, K, G, pt, grid, flow_bcs, dt).  Call init_global_grid first for N > 1."""
    import torch
    from ..arrays import StokesArrays, fzeros
    from ..backend import device_of
    from ..grid import grid_is_initialized
    ni = (n, n, n) if isinstance(n, int) else tuple(n)
    if not grid_is_initialized():
        init_global_grid(*ni)
    dev = device_of(backend_tag)
    di = tuple(l / g for l, g in zip(li, (nx_g(), ny_g(), nz_g())))
    grid = Geometry(ni, li, origin=(0.0, 0.0, 0.0))
    pt = PTStokesCoeffs(li, di, CFL=1 / math.sqrt(3))
    stokes = StokesArrays(backend_tag, ni)
    gg = global_grid()
    off = [gg.coords[d] * (ni[d] - OVERLAP) for d in range(3)]
    ax = [(torch.arange(ni[d], device=dev, dtype=torch.float64) + off[d]) * di[d] + 0.5 * di[d] - 0.5 * li[d] for d in range(3)]
    η, η2 = stokes.viscosity.η, fzeros(ni, dev)
    r2 = ax[0][:, None, None] ** 2 + ax[1][None, :, None] ** 2 + ax[2][None, None, :] ** 2
    η.fill_(1.0)
    η[torch.sqrt(r2) <= rc] = Δη
    del r2
    η2.copy_(η)
    a, b = η, η2
    for _ in range(10):            # 10 even ping-pong passes: the result ends in `a` == stokes.viscosity.η
        c = a[1:-1, 1:-1, 1:-1]
        lap = ((a[2:, 1:-1, 1:-1] - c) - (c - a[:-2, 1:-1, 1:-1])) + ((a[1:-1, 2:, 1:-1] - c) - (c - a[1:-1, :-2, 1:-1]))
        lap = lap + ((a[1:-1, 1:-1, 2:] - c) - (c - a[1:-1, 1:-1, :-2]))
        b[1:-1, 1:-1, 1:-1] = c + 1.0 / 6.1 / 1.0 * lap
        del lap
        a, b = b, a
        if update_halo is not None:
            update_halo(a)
    assert a is η
    del η2
    G = fzeros(ni, dev, 1.0)
    K = fzeros(ni, dev, float("inf"))
    ρg = tuple(fzeros(ni, dev) for _ in range(3))
    xv = torch.as_tensor(grid.xvi[0], device=dev)
    zv = torch.as_tensor(grid.xvi[2], device=dev)
    stokes.V.Vx[:, 1:-1, 1:-1] = (εbg * xv)[:, None, None]
    stokes.V.Vy[1:-1, :, 1:-1] = (εbg * xv)[None, :, None]
    stokes.V.Vz[1:-1, 1:-1, :] = (-εbg * zv)[None, None, :]
    bcs = VelocityBoundaryConditions(free_slip={f: True for f in _F6}, no_slip={f: False for f in _F6})
    return stokes, ρg, K, G, pt, grid, bcs, float("inf")
