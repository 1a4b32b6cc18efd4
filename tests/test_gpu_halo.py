"""GPU tests of update_halo! (ImplicitGlobalGrid semantics) on one device.

A 1-GPU box has no neighbour rank, so the exchange is exercised through periodic dimensions held by the rank
itself (IGG copies locally in that case): once through the library's local-copy path and once, with the test hook
JRX_HALO_SELF_RCCL=1, through a one-rank RCCL communicator -- the same pack kernel -> grouped ncclSend/ncclRecv ->
unpack kernel sequence that carries the planes between GPUs (src/stokes/Stokes3D.jl:57,120 call sites).
The expectation is computed in numpy from jrx_halo_planes (x, then y, then z).  Bit-exact.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expected(arrs, n, periods, L):
    out = [a.copy(order="F") for a in arrs]
    for dim in range(3):
        if not periods[dim]:
            continue
        for A in out:
            sl, sr, rl, rr = (C.c_int64() for _ in range(4))
            if L.jrx_halo_planes(C.c_int64(n[dim]), C.c_int64(A.shape[dim]), C.byref(sl), C.byref(sr), C.byref(rl), C.byref(rr)) != 0:
                continue
            left_going = np.take(A, sl.value, axis=dim).copy()      # my left send plane arrives in (my own) right ghost plane
            right_going = np.take(A, sr.value, axis=dim).copy()
            idx = [slice(None)] * 3
            idx[dim] = rr.value
            A[tuple(idx)] = left_going
            idx[dim] = rl.value
            A[tuple(idx)] = right_going
    return out


@pytest.mark.parametrize("through_rccl", [False, True])
@pytest.mark.parametrize("periods", [(1, 0, 0), (0, 1, 1), (1, 1, 1)])
def test_periodic_self_exchange(jr, through_rccl, periods):
    import torch
    from justrelax_jl_amd import _lib, halo
    from justrelax_jl_amd.arrays import from_numpy, to_numpy
    import justrelax_jl_amd.grid as g
    L = _lib.load()
    n = (21, 12, 9)
    rng = np.random.default_rng(3)
    shapes = [(n[0] + 1, n[1] + 2, n[2] + 2), (n[0] + 2, n[1] + 1, n[2] + 2), (n[0] + 2, n[1] + 2, n[2] + 1), n]
    host = [np.asfortranarray(rng.standard_normal(s)) for s in shapes]
    g.init_global_grid(*n, periodx=periods[0], periody=periods[1], periodz=periods[2], rank=0, nprocs=1)
    old = os.environ.get("JRX_HALO_SELF_RCCL")
    h = _lib.Handle(torch.cuda.current_device())
    try:
        if through_rccl:
            os.environ["JRX_HALO_SELF_RCCL"] = "1"
        halo.init_comm(h)
        dev = [from_numpy(a, torch.device('cuda', torch.cuda.current_device())) for a in host]
        halo.update_halo_(*dev, ni=n, handle=h)
        torch.cuda.synchronize()
        got = [to_numpy(d) for d in dev]
    finally:
        if old is None:
            os.environ.pop("JRX_HALO_SELF_RCCL", None)
        else:
            os.environ["JRX_HALO_SELF_RCCL"] = old
        h.close()
        g.finalize_global_grid()
    exp = _expected(host, n, periods, L)
    for a, b, s in zip(got, exp, shapes):
        assert np.array_equal(a, b), f"halo mismatch for array of shape {s}, periods {periods}, rccl={through_rccl}"
        # interior untouched
    assert any(not np.array_equal(a, b) for a, b in zip(got, host))
