"""Worker for tests/test_multirank_gloo.py (world_size 2, gloo, CPU only).

Exercises the N > 1 host logic of the product -- block decomposition (grid.init_global_grid /
jrx_cart_create), halo plane selection (jrx_halo_planes) and the x -> y -> z exchange order -- with
the CPU oracle as the compute kernel and gloo as the transport (on GPUs the same plan drives
pack kernels + RCCL inside libjrx_hip).  Checks, on every rank:
  1. after update_halo the ghost planes hold the neighbour's send planes;
  2. a decomposed PT run with uniform material equals the undecomposed run bit for bit on every
     cell the rank owns (IGG overlap semantics: duplicated overlap cells stay consistent);
  3. the summed Σx² norms double-count the overlap exactly as the reference's norm_mpi does.
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
    assert tuple(gg.dims) == (2, 1, 1) and g.nx_g() == 2 * (n[0] - 2) + 2 == 18
    cart = halo.make_cart(gg)
    assert cart.neighbor[0][0] == (-1 if rank == 0 else 0) and cart.neighbor[0][1] == (1 if rank == 0 else -1)
    ng = (g.nx_g(), g.ny_g(), g.nz_g())

    # global problem (identical on both ranks), uniform material so that the clamped shear averages
    # at rank-internal faces equal the global ones
    g.finalize_global_grid()
    S = jr.miniapps.random_fields3d(ng, seed=5, iterMax=12, nout=4)
    S.pt.ϵ_rel = S.pt.ϵ_abs = 1e-30
    for k in ("eta", "G", "K"):
        S.arrays[k][...] = {"eta": 0.7, "G": 1.3, "K": 2.1}[k]
    g.init_global_grid(*n, rank=rank, nprocs=world)
    off = rank * (n[0] - 2)

    def local(name, A):
        return np.asfortranarray(A[off: off + A.shape[0] - ng[0] + n[0]])

    loc = {k: local(k, v) for k, v in S.arrays.items()}
    shp = orc.shapes3d(*n)
    for k, v in loc.items():
        assert v.shape == shp[k], (k, v.shape, shp[k])

    # 1. exchange check on a copy
    V = [loc[k].copy(order="F") for k in ("Vx", "Vy", "Vz")]
    for A in V:
        A[0], A[-1] = -777.0, -777.0
    update_halo_gloo(V, n, cart, L)
    for A, k in zip(V, ("Vx", "Vy", "Vz")):
        if rank == 1:
            assert np.array_equal(A[0], loc[k][0]) and (A[-1] == -777.0).all()
        else:
            assert np.array_equal(A[-1], loc[k][-1]) and (A[0] == -777.0).all()

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
        o = r * (n[0] - 2)
        lr = {k: np.asfortranarray(glob[k][o: o + glob[k].shape[0] - ng[0] + n[0]]) for k in ("Rx", "Ry", "Rz", "RP")}
        full = dict(loc)
        full.update(lr)
        want += orc.residual_sumsq3d(full, pl)
    assert np.allclose(sums[-1], want, rtol=1e-12), (sums[-1], want)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
