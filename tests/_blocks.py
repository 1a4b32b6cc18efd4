"""Host-side helpers of the two-block (N > 1) tests: cut the blocks of an ImplicitGlobalGrid decomposition out of global arrays and
replay update_halo! between them in numpy.

Semantics restated from the reference's call sites (src/stokes/Stokes3D.jl:57,117-120; src/grid/Utils.jl:26-39) and SURVEY §5: local
blocks of n cells overlap their neighbours by 2 cells, rank offset coords * (n - 2); an array of extent nA exchanges the planes
jrx_halo_planes names, x first, then y, then z.
"""
import ctypes as C

import numpy as np


def coords_of(cart):
    return tuple(cart.coords[d] for d in range(3))


def local_block(A, n, ng, coords, nd=3):
    """the block of rank `coords` of a global array A (any staggering): extent along d = A.shape[d] - ng[d] + n[d]; nd = 2 for the 2D drivers"""
    lead = A.ndim - nd          # phase-ratio arrays carry the phase index first
    idx = [slice(None)] * lead
    for d in range(nd):
        off = coords[d] * (n[d] - 2)
        idx.append(slice(off, off + A.shape[lead + d] - ng[d] + n[d]))
    return np.array(A[tuple(idx)], order="F", copy=True)      # always a copy: a z slab of an F-ordered array is contiguous, asfortranarray would alias it


def n_global(n, dims, periods=(0, 0, 0)):
    return tuple(dims[d] * (n[d] - 2) + (0 if periods[d] else 2) if n[d] > 1 else 1 for d in range(3))


def exchange(blocks, n, carts, L):
    """update_halo!(A...) between the blocks of one process: blocks[r] = list of that rank's arrays (same order on every rank), in place."""
    nr = len(blocks)
    for dim in range(blocks[0][0].ndim):
        if all(carts[r].neighbor[dim][0] < 0 and carts[r].neighbor[dim][1] < 0 for r in range(nr)):
            continue
        for a in range(len(blocks[0])):
            ext = blocks[0][a].shape
            sl, sr, rl, rr = (C.c_int64() for _ in range(4))
            if L.jrx_halo_planes(C.c_int64(n[dim]), C.c_int64(ext[dim]), C.byref(sl), C.byref(sr), C.byref(rl), C.byref(rr)) != 0:
                continue
            send = [(np.take(blocks[r][a], sl.value, axis=dim).copy(), np.take(blocks[r][a], sr.value, axis=dim).copy()) for r in range(nr)]
            for r in range(nr):
                for side, rp in ((0, rl.value), (1, rr.value)):
                    nb = carts[r].neighbor[dim][side]
                    if nb < 0:
                        continue
                    idx = [slice(None)] * blocks[r][a].ndim
                    idx[dim] = rp
                    blocks[r][a][tuple(idx)] = send[nb][1 - side]     # my left ghost plane <- the left neighbour's right-going plane


def owned_mask(shape, n, cart, name=None):
    """entries of a local array that are not duplicates of a neighbour's interior: everything except the outermost plane on a face
    with a neighbour (those planes are received, or are the overlap copy the neighbour computes with a full stencil)"""
    m = np.ones(shape, dtype=bool)
    for d in range(3):
        for side in (0, 1):
            if cart.neighbor[d][side] >= 0:
                idx = [slice(None)] * 3
                idx[d] = 0 if side == 0 else shape[d] - 1
                m[tuple(idx)] = False
    return m
