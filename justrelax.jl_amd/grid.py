"""Uniform Cartesian staggered grid and the implicit global grid (block decomposition).

Reference: src/grid/Cartesian.jl:9-58 (Geometry), src/grid/Grid.jl:18-24 (IGG), :56-143
(geometry_MPI / lazy_grid), src/grid/Utils.jl:20-60 (x_g), and the semantics of the un-vendored
ImplicitGlobalGrid.jl 0.16/0.17 that the reference relies on: local arrays of `nx`
cells overlap their neighbours by `ol = 2` cells, `nx_g = dims*(nx - ol) + ol`, rank offset
`coord*(nx - ol)`.

One process per GPU; the process group comes from torch.distributed (RCCL on GPUs, gloo in the
CPU tests).  Nothing here touches device memory.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

OVERLAP = 2   # ImplicitGlobalGrid default overlaps (2,2,2)


@dataclass
class _GlobalGrid:
    nxyz: tuple = (1, 1, 1)
    dims: tuple = (1, 1, 1)
    coords: tuple = (0, 0, 0)
    periods: tuple = (0, 0, 0)
    me: int = 0
    nprocs: int = 1
    initialized: bool = False

    def n_g(self, d):
        n, D = self.nxyz[d], self.dims[d]
        if n == 1:
            return 1
        return D * (n - OVERLAP) + (0 if self.periods[d] else OVERLAP)


_GG = _GlobalGrid()


def global_grid() -> _GlobalGrid:
    return _GG


def grid_is_initialized() -> bool:
    return _GG.initialized


def dims_create(nprocs: int, active: tuple) -> tuple:
    """Balanced Cartesian factorisation (what MPI_Dims_create gives IGG): largest factors first,
    only over dimensions with more than one cell.  8 -> (2,2,2), 4 -> (2,2,1), 2 -> (2,1,1)."""
    nd = sum(active)
    dims = [1] * 3
    if nd == 0 or nprocs == 1:
        return tuple(dims)
    fac, n, p = [], nprocs, 2
    while n > 1:
        while n % p == 0:
            fac.append(p)
            n //= p
        p += 1
    idx = [d for d in range(3) if active[d]]
    for f in sorted(fac, reverse=True):
        j = min(idx, key=lambda d: dims[d])
        dims[j] *= f
    vals = sorted((dims[d] for d in idx), reverse=True)
    for d, v in zip(idx, vals):
        dims[d] = v
    return tuple(dims)


def cart_coords(rank: int, dims) -> tuple:
    """Row-major Cartesian rank -> coords (MPI_Cart_coords convention: last dim fastest)."""
    c = [0, 0, 0]
    for d in (2, 1, 0):
        c[d] = rank % dims[d]
        rank //= dims[d]
    return tuple(c)


def cart_rank(coords, dims) -> int:
    r = 0
    for d in range(3):
        r = r * dims[d] + coords[d]
    return r


def neighbors(coords, dims, periods=(0, 0, 0)):
    """[(left, right)] per dimension; -1 where the face is a physical boundary."""
    out = []
    for d in range(3):
        pair = []
        for s in (-1, +1):
            c = list(coords)
            c[d] += s
            if 0 <= c[d] < dims[d]:
                pair.append(cart_rank(c, dims))
            elif periods[d]:
                c[d] %= dims[d]
                pair.append(cart_rank(c, dims))
            else:
                pair.append(-1)
        out.append(tuple(pair))
    return out


def init_global_grid(nx, ny, nz=1, *, dimx=0, dimy=0, dimz=0, periodx=0, periody=0, periodz=0,
                     init_MPI=True, rank=None, nprocs=None, quiet=True):
    """ImplicitGlobalGrid.init_global_grid as the miniapps call it
    (e.g. miniapps/benchmarks/stokes3D/solvi/SolVi3D.jl:63).  Returns (me, dims, nprocs, coords, comm)."""
    if rank is None or nprocs is None:
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank, nprocs = dist.get_rank(), dist.get_world_size()
        except Exception:
            pass
    if rank is None or nprocs is None:
        rank, nprocs = int(os.environ.get("RANK", 0)), 1
    active = (nx > 1, ny > 1, nz > 1)
    fixed = (dimx, dimy, dimz)
    if any(fixed):
        dims = tuple(f if f else 1 for f in fixed)
        if int(np.prod(dims)) != nprocs:
            raise ValueError(f"dims {dims} do not multiply to the number of processes {nprocs}")
    else:
        dims = dims_create(nprocs, active)
    _GG.nxyz, _GG.dims, _GG.me, _GG.nprocs = (nx, ny, nz), dims, rank, nprocs
    _GG.coords = cart_coords(rank, dims)
    _GG.periods = (periodx, periody, periodz)
    _GG.initialized = True
    return rank, list(dims), nprocs, list(_GG.coords), None


def finalize_global_grid(**_):
    _GG.initialized = False
    _GG.dims, _GG.coords, _GG.me, _GG.nprocs = (1, 1, 1), (0, 0, 0), 0, 1


def nx_g():
    return _GG.n_g(0)


def ny_g():
    return _GG.n_g(1)


def nz_g():
    return _GG.n_g(2)


def x_g(idx, dxi, nxi, d=0):
    """src/grid/Utils.jl:24-40 (idx is 1-based as in the reference)."""
    n = _GG.nxyz[d]
    x0i = 0.5 * (n - nxi) * dxi
    xi = (_GG.coords[d] * (n - OVERLAP) + idx - 1) * dxi + x0i
    if _GG.periods[d]:
        ng = _GG.n_g(d)
        xi = xi - dxi
        if xi > (ng - 1) * dxi:
            xi -= ng * dxi
        if xi < 0:
            xi += ng * dxi
    return xi


@dataclass
class IGG:
    """src/grid/Grid.jl:18-24"""
    me: int = 0
    dims: list = field(default_factory=lambda: [1, 1, 1])
    nprocs: int = 1
    coords: list = field(default_factory=lambda: [0, 0, 0])
    comm_cart: object = None


class Geometry:
    """Geometry(ni, li; origin) -- src/grid/Cartesian.jl:42-58, src/grid/Grid.jl:56-143; Geometry.from_vertices(xvi) is the non-uniform constructor
    Geometry(xvi::NTuple) (Cartesian.jl:77-100)."""

    nonuniform = False

    @classmethod
    def from_vertices(cls, xvi):
        """Geometry(TA, xvi...) / Geometry(xvi::NTuple) -- src/grid/Cartesian.jl:77-100: a staggered grid from explicit vertex coordinates; spacings are
        vectors: di.vertex = diff(xvi) (n), di.center = diff(xci) (n - 1), di.velocity[i][d] = diff of the ghosted velocity grids (Grid.jl:171-215)"""
        xvi = tuple(np.asarray(x, dtype=np.float64) for x in xvi)
        nD = len(xvi)
        g = cls.__new__(cls)
        g.nonuniform = True
        g.ni = tuple(len(x) - 1 for x in xvi)
        g.xvi = xvi
        g.xci = tuple((x[:-1] + x[1:]) / 2 for x in xvi)
        g.li = tuple(float(x.max() - x.min()) for x in xvi)
        g.origin = tuple(float(x.min()) for x in xvi)
        g.max_li = max(g.li)
        dv, dc = tuple(np.diff(x) for x in xvi), tuple(np.diff(x) for x in g.xci)
        ghost = tuple(np.concatenate(([c[0] - d[0]], c, [c[-1] + d[-1]])) for c, d in zip(g.xci, dc))      # velocity_grids, Grid.jl:171-182,202-215
        g.xi_vel = tuple(tuple(xvi[d] if d == i else ghost[d] for d in range(nD)) for i in range(nD))
        dvel = tuple(tuple(np.diff(x) for x in gv) for gv in g.xi_vel)
        g.di = dict(center=dc, vertex=dv, velocity=dvel)
        g._di = dict(center=tuple(1.0 / x for x in dc), vertex=tuple(1.0 / x for x in dv), velocity=tuple(tuple(1.0 / x for x in gv) for gv in dvel))
        g._dev = {}
        return g

    def inv_spacing2d(self, device):
        """the six inverse-spacing arrays of the 2D C ABI (jrx_stokes2d_params.inv_spacing) as device tensors, cached per device"""
        import torch
        if device not in self._dev:
            d = self._di
            host = (d["vertex"][0], d["vertex"][1], d["center"][0], d["center"][1], d["velocity"][0][1], d["velocity"][1][0])
            self._dev[device] = tuple(torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=device) for a in host)
        return self._dev[device]

    def inv_spacing2d_host(self):
        d = self._di
        return (d["vertex"][0], d["vertex"][1], d["center"][0], d["center"][1], d["velocity"][0][1], d["velocity"][1][0])

    def __init__(self, ni, li, origin=None):
        nD = len(ni)
        origin = tuple(float(o) for o in (origin if origin is not None else (0.0,) * nD))
        self.ni = tuple(int(n) for n in ni)
        self.li = tuple(float(l) for l in li)
        self.origin = origin
        self.max_li = max(self.li)
        if grid_is_initialized():                       # geometry_MPI
            ni_g = tuple(_GG.n_g(d) for d in range(nD))
            di = tuple(l / n for l, n in zip(self.li, ni_g))
            xci, xvi = [], []
            for d in range(nD):
                o0 = x_g(1, di[d], self.ni[d], d) + origin[d]
                xci.append(np.linspace(o0 + di[d] / 2, x_g(self.ni[d], di[d], self.ni[d], d) + origin[d] + di[d] / 2, self.ni[d]))
                xvi.append(np.linspace(o0, x_g(self.ni[d] + 1, di[d], self.ni[d], d) + origin[d], self.ni[d] + 1))
        else:                                           # geometry_nonMPI
            di = tuple(l / n for l, n in zip(self.li, self.ni))
            xci = [np.linspace(origin[d] + di[d] / 2, origin[d] + self.li[d] - di[d] / 2, self.ni[d]) for d in range(nD)]
            xvi = [np.linspace(origin[d], origin[d] + self.li[d], self.ni[d] + 1) for d in range(nD)]
        self.xci, self.xvi = tuple(xci), tuple(xvi)
        # velocity_grids (Grid.jl:161-169,184-200): the grid of V_i is the vertices along i and the centres, with one ghost point either side, along the others
        ghost = tuple(np.linspace(xci[d][0] - di[d], xci[d][-1] + di[d], self.ni[d] + 2) for d in range(nD))
        self.xi_vel = tuple(tuple(self.xvi[d] if d == i else ghost[d] for d in range(nD)) for i in range(nD))
        inv = tuple(1.0 / d for d in di)
        self.di = dict(center=di, vertex=di, velocity=tuple(di for _ in range(nD)))
        self._di = dict(center=inv, vertex=inv, velocity=tuple(inv for _ in range(nD)))


def legacy_uniform_grid(ni, di) -> Geometry:
    """src/grid/Grid.jl:41-51"""
    if isinstance(di, dict):
        di = di["center"]
    nD = len(ni)
    ni_g = tuple(_GG.n_g(d) for d in range(nD)) if grid_is_initialized() else tuple(ni)
    return Geometry(ni, tuple(float(di[d]) * ni_g[d] for d in range(nD)))
