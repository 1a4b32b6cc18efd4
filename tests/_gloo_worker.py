"""Worker for tests/test_multirank_gloo.py (world_size 2 -> (2,1,1) and world_size 8 -> (2,2,2), gloo, CPU only).

Exercises the N > 1 host logic of the product -- block decomposition (grid.init_global_grid /
jrx_cart_create), halo plane selection (jrx_halo_planes) and the x -> y -> z exchange order -- with
the CPU oracle as the compute kernel and gloo as the transport (on GPUs the same plan drives
pack kernels + RCCL inside libjrx_hip).  Checks, on every rank:
  1. after update_halo the ghost planes hold the neighbour's send planes;
  2. a decomposed PT run with uniform material equals the undecomposed run bit for bit on every
     cell the rank owns (IGG overlap semantics: duplicated overlap cells stay consistent);
  3. the summed Σx² norms double-count the overlap exactly as the reference's norm_mpi does
     (Stokes3D.jl:127-142, nx_g = dims_x (n - 2) + 2 in every split dimension).
With 8 ranks the x -> y -> z order is what carries edge and corner ghosts: a rank's corner entry comes from its diagonal
neighbour in three hops, and check 2 fails on the first iteration if one of them is out of order.
"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def update_halo_gloo(arrays, n, cart, L):
    """update_halo!(A...) with gloo send/recv; plane indices and neighbours come from the C ABI."""
    for dim in range(3):
        left, right = cart.neighbor[dim][0], cart.neighbor[dim][1]
        if left < 0 and right < 0:
            continue
        reqs, recvs = [], []
        for A in arrays:
            ext = A.shape + (1,) * (3 - A.ndim)
            sl, sr, rl, rr = (C.c_int64() for _ in range(4))
            if L.jrx_halo_planes(C.c_int64(n[dim]), C.c_int64(ext[dim]), C.byref(sl), C.byref(sr), C.byref(rl), C.byref(rr)) != 0:
                continue
            A3 = A.reshape(ext, order="F")
            take = lambda p: torch.from_numpy(np.ascontiguousarray(np.take(A3, p, axis=dim)))
            for nb, sp, rp in ((left, sl.value, rl.value), (right, sr.value, rr.value)):
                if nb < 0:
                    continue
                reqs.append(dist.isend(take(sp), nb))
                buf = torch.empty_like(take(rp))
                reqs.append(dist.irecv(buf, nb))
                recvs.append((A3, rp, buf))
        for r in reqs:
            r.wait()
        for A3, rp, buf in recvs:
            idx = [slice(None)] * 3
            idx[dim] = rp
            A3[tuple(idx)] = buf.numpy()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    jr = load_package()
    import oracle as orc
    from justrelax_jl_amd import _lib, checks, halo
    import justrelax_jl_amd.grid as g
    L = _lib.load()

    n = (10, 9, 8)
    g.init_global_grid(*n, rank=rank, nprocs=world)
    gg = g.global_grid()
    dims = tuple(gg.dims)
    assert dims == {2: (2, 1, 1), 8: (2, 2, 2)}[world], dims
    ng = (g.nx_g(), g.ny_g(), g.nz_g())
    assert ng == tuple(d * (m - 2) + 2 for d, m in zip(dims, n)), ng
    cart = halo.make_cart(gg)
    coords = tuple(gg.coords)
    rank_of = lambda c: g.cart_rank(c, dims)          # MPI_Cart_rank order (last dimension fastest), as IGG has it
    assert coords == tuple(g.cart_coords(rank, dims))
    for d in range(3):
        for side, step in ((0, -1), (1, 1)):
            c = list(coords); c[d] += step
            want = rank_of(c) if 0 <= c[d] < dims[d] else -1
            assert cart.neighbor[d][side] == want, (rank, d, side, cart.neighbor[d][side], want)

    # global problem (identical on all ranks), uniform material so that the clamped shear averages
    # at rank-internal faces equal the global ones
    g.finalize_global_grid()
    S = jr.miniapps.random_fields3d(ng, seed=5, iterMax=12, nout=4)
    S.pt.ϵ_rel = S.pt.ϵ_abs = 1e-30
    for k in ("eta", "G", "K"):
        S.arrays[k][...] = {"eta": 0.7, "G": 1.3, "K": 2.1}[k]
    g.init_global_grid(*n, rank=rank, nprocs=world)

    def block(A, c):
        """the local block of rank coordinates c out of the global array A (any staggering: the local extent is the global one less ng - n)"""
        sl = tuple(slice(c[d] * (n[d] - 2), c[d] * (n[d] - 2) + A.shape[d] - ng[d] + n[d]) for d in range(3))
        return np.asfortranarray(A[sl])

    local = lambda name, A: block(A, coords)

    loc = {k: local(k, v) for k, v in S.arrays.items()}
    shp = orc.shapes3d(*n)
    for k, v in loc.items():
        assert v.shape == shp[k], (k, v.shape, shp[k])

    # 1. exchange check on a copy: ghost planes poisoned, then update_halo -- faces with a neighbour hold the neighbour's values (= the global field there, edges and
    #    corners included when all three dimensions are split), faces without one keep the poison
    V = [loc[k].copy(order="F") for k in ("Vx", "Vy", "Vz")]
    for A in V:
        for d in range(3):
            idx = [slice(None)] * 3
            for p in (0, -1):
                idx[d] = p
                A[tuple(idx)] = -777.0
    update_halo_gloo(V, n, cart, L)
    for A, k in zip(V, ("Vx", "Vy", "Vz")):
        inner = tuple(slice(1 if cart.neighbor[d][0] < 0 else 0, -1 if cart.neighbor[d][1] < 0 else None) for d in range(3))
        assert np.array_equal(A[inner], loc[k][inner]), (rank, k)
        for d in range(3):
            for side, p in ((0, 0), (1, -1)):
                if cart.neighbor[d][side] < 0:
                    idx = [slice(None)] * 3
                    idx[d] = p
                    assert (A[tuple(idx)] == -777.0).all(), (rank, k, d, side)

    # 2. decomposed run == global run
    b = S.flow_bcs
    pl = orc.params3d(n, S.grid._di["center"], S.dt, dict(r=S.pt.r, theta_dtau=S.pt.θ_dτ, eta_dtau=S.pt.ηdτ, eps_rel=1e-30, eps_abs=1e-30),
                      iterMax=12, nout=4, free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, ni_g=ng)
    et = orc.compute_maxloc(loc["eta"])
    update_halo_gloo([et], n, cart, L)
    sums = []
    for it in range(1, 9):
        orc.stokes3d_iteration(loc, et, pl)
        update_halo_gloo([loc["Vx"], loc["Vy"], loc["Vz"]], n, cart, L)
        if it % 4 == 0:
            s = torch.from_numpy(orc.residual_sumsq3d(loc, pl))
            dist.all_reduce(s)
            sums.append(s.numpy().copy())
    glob = {k: v.copy(order="F") for k, v in S.arrays.items()}
    pg = checks.oracle_params3d(orc, S)
    etg = orc.compute_maxloc(glob["eta"])
    for it in range(1, 9):
        orc.stokes3d_iteration(glob, etg, pg)
    for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "txy", "txz", "tyz"):
        want = local(k, glob[k])
        m = checks.interior_mask3d(k, want.shape)
        assert np.array_equal(loc[k][m], want[m]), (rank, k, np.abs(loc[k] - want).max())

    # 3. norm_mpi double counts the 2-cell overlap (Stokes3D.jl:127-142): Σ over ranks of local interior slices
    want = np.zeros(4)
    for r in range(world):
        c = tuple(g.cart_coords(r, dims))
        lr = {k: block(glob[k], c) for k in ("Rx", "Ry", "Rz", "RP")}
        full = dict(loc)
        full.update(lr)
        want += orc.residual_sumsq3d(full, pl)
    assert np.allclose(sums[-1], want, rtol=1e-12), (sums[-1], want)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
