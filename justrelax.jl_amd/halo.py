"""Multi-GPU plumbing: one process per GPU, RCCL communicator inside the native handle.

update_halo_ mirrors ImplicitGlobalGrid.update_halo! at the reference's call sites
(src/stokes/Stokes3D.jl:57,120 ...).  torch.distributed is used only to ship the 128-byte RCCL
unique id from rank 0 to the other ranks (what MPI.Bcast would do in the Julia shim).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .arrays import ptr
from .grid import global_grid


def make_cart(gg=None) -> _lib.Cart:
    """jrx_cart_create from the current global grid (host logic only; works without a GPU)."""
    gg = gg or global_grid()
    L = _lib.load()
    cart = _lib.Cart()
    n = (C.c_int64 * 3)(*gg.nxyz)
    dims = (C.c_int32 * 3)(*gg.dims)
    per = (C.c_int32 * 3)(*gg.periods)
    st = L.jrx_cart_create(C.c_int32(gg.me), C.c_int32(gg.nprocs), n, dims, per, C.byref(cart))
    if st != 0:
        raise _lib.JrxError(st, "jrx_cart_create failed")
    return cart


def init_comm(handle=None, group=None, self_rccl=False):
    """Create the RCCL communicator of `handle` for the ranks of the torch.distributed group.
    `self_rccl` (test hook, one rank): a periodic dimension held by the rank alone is exchanged through ncclSend/ncclRecv on a
    one-rank communicator instead of the local copy."""
    import torch.distributed as dist
    h = handle or _lib.default_handle()
    gg = global_grid()
    cart = make_cart(gg)
    h.call("jrx_tuning_set", C.c_char_p(b"halo_self_rccl"), C.c_int64(int(bool(self_rccl))))
    if gg.nprocs == 1 and not self_rccl:
        # a periodic dimension held by one rank is exchanged by a local copy inside the library
        h.call("jrx_comm_init", None, C.byref(cart))
        return h
    uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
    if gg.me == 0:
        st = h.lib.jrx_comm_unique_id(uid)
        if st != 0:
            raise _lib.JrxError(st, h.lib.jrx_last_error(None).decode())
    if gg.nprocs > 1:
        obj = [bytes(uid) if gg.me == 0 else None]
        dist.broadcast_object_list(obj, src=0, group=group)
        uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(obj[0])
    h.call("jrx_comm_init", uid, C.byref(cart))
    return h


def init_comm_ipc(handle=None, group=None, cart=None):
    """jrx_comm_init_ipc: one process per rank on one node, packed planes pushed into the neighbour process's receive buffer by copy engines (IPC memory
    handles), ordered by flags in a shared-memory segment.  The 128 bytes that name the segment travel like the RCCL id (torch.distributed here, MPI.Bcast
    in the Julia extension).  `cart`: defaults to the current global grid's."""
    import torch.distributed as dist
    h = handle or _lib.default_handle()
    cart = cart or make_cart(global_grid())
    uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
    if cart.nprocs > 1:
        if cart.rank == 0:
            st = h.lib.jrx_comm_ipc_id(uid)
            if st != 0:
                raise _lib.JrxError(st, "jrx_comm_ipc_id failed")
        obj = [bytes(uid) if cart.rank == 0 else None]
        dist.broadcast_object_list(obj, src=0, group=group)
        uid = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(obj[0])
    h.call("jrx_comm_init_ipc", uid, C.byref(cart))
    return h


def make_carts(n, dims, periods=(0, 0, 0)):
    """The carts of every rank of a `dims` process grid of local blocks of `n` cells (host logic only)."""
    L = _lib.load()
    nprocs = int(np.prod(dims))
    carts = (_lib.Cart * nprocs)()
    for r in range(nprocs):
        st = L.jrx_cart_create(C.c_int32(r), C.c_int32(nprocs), (C.c_int64 * 3)(*n), (C.c_int32 * 3)(*dims), (C.c_int32 * 3)(*periods), C.byref(carts[r]))
        if st != 0:
            raise _lib.JrxError(st, "jrx_cart_create failed")
    return carts


def init_comm_local(handles, carts):
    """jrx_comm_init_local: the ranks are `handles` of this process (one device or peer-accessible devices); planes travel by
    device-to-device copies.  Every rank must then be driven by its own host thread (`run_ranks`)."""
    n = len(handles)
    arr = (C.c_void_p * n)(*[h._h for h in handles])
    st = handles[0].lib.jrx_comm_init_local(arr, C.c_int32(n), carts)
    handles[0].check(st)
    return handles


def run_ranks(fns):
    """Run one callable per rank, each on its own host thread (ctypes releases the GIL inside the library), and return their results;
    the first exception of any rank is re-raised."""
    import threading
    out, err = [None] * len(fns), [None] * len(fns)

    def work(r):
        try:
            out[r] = fns[r]()
        except BaseException as e:       # noqa: BLE001 -- re-raised below
            err[r] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out


def update_halo_(*fields, ni=None, handle=None):
    """update_halo!(A...) on device arrays (up to 8 per call); `ni` = local cell counts (defaults to the global grid's)."""
    h = handle or _lib.default_handle(fields[0].device.index)
    gg = global_grid()
    ni = tuple(ni or gg.nxyz)
    ni = ni + (1,) * (3 - len(ni))
    na = len(fields)
    arrs = (C.c_void_p * na)(*[ptr(f) for f in fields])
    ext = ((C.c_int64 * 3) * na)()
    for a, f in enumerate(fields):
        shp = tuple(f.shape) + (1,) * (3 - f.dim())
        for d in range(3):
            ext[a][d] = shp[d]
    n = (C.c_int64 * 3)(*ni)
    torch.cuda.current_stream(fields[0].device).synchronize()
    h.call("jrx_update_halo", C.c_int32(na), arrs, ext, n)
