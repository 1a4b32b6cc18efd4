"""jrx_field_alloc / jrx_field_free: the library hands out the state arrays (the backend owns the array constructor in the reference:
src/ext/AMDGPU/3D.jl:46-48 StokesArrays(::Type{AMDGPUBackend}, ni) -> @zeros(ni...)).  Whatever "field_placement" selects -- hipMalloc,
physical chunks dealt at random from a pool and mapped once at a fresh virtual range, physically contiguous memory -- an array is plain device memory: a solve on library arrays gives
the bits of the same solve on torch's arrays, and the pool accounts for every array it handed out."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stats(h):
    s = (C.c_int64 * 6)()
    h.call("jrx_field_stats", s)
    return list(s)


@pytest.mark.parametrize("placement,chunk", [(0, 64), (1, 2), (1, 8), (2, 64)])
def test_alloc_write_read_free(jr, placement, chunk):
    import torch
    from justrelax_jl_amd import _lib, arrays
    h = _lib.Handle(0)
    try:
        h.set_option("field_placement", placement)
        h.set_option("field_chunk_mib", chunk)
        arrays.use_library_arrays(h)
        shapes = [(257, 130, 67), (3, 5, 7), (1200, 1100), (64, 64, 64)]        # 17.9 MB, tiny (always hipMalloc), 10.6 MB, 2 MB
        ts = [jr.fzeros(s, "cuda", fill=float(i + 1)) for i, s in enumerate(shapes)]
        st = _stats(h)
        assert st[0] == len(shapes) and st[1] == sum(int(np.prod(s)) * 8 for s in shapes)
        if placement == 1:
            assert st[2] >= sum(-(-int(np.prod(s)) * 8 // (chunk << 20)) for s in shapes if int(np.prod(s)) * 8 >= (8 << 20))
        for i, (t, s) in enumerate(zip(ts, shapes)):
            assert tuple(t.shape) == s and arrays.is_fortran(t)
            assert float(t.min()) == float(t.max()) == float(i + 1)
        # the last element of every array is addressable and the arrays do not overlap
        rng = np.random.default_rng(7)
        ref = []
        for t in ts:
            a = rng.standard_normal(tuple(t.shape))
            t.copy_(torch.from_numpy(a).to("cuda"))
            ref.append(a)
        for t, a in zip(ts, ref):
            assert np.array_equal(t.cpu().numpy(), a)
        p0 = ts[0].data_ptr()
        del ts, t
        torch.cuda.synchronize()
        assert _stats(h)[0] == 0 and _stats(h)[1] == 0
        # chunk sizes without a pool: the memory of a freed array went back to the driver, and its virtual range is never handed out again
        assert _stats(h)[3] == 0
        t = jr.fzeros(shapes[0], "cuda")
        assert p0 != 0 and t.data_ptr() != 0 and (placement != 1 or t.data_ptr() != p0)
        del t
        h.call("jrx_field_trim")
        assert _stats(h)[3] == 0
    finally:
        arrays.use_library_arrays(None)
        h.close()


def test_pool_placement_maps_every_array_once_at_a_range_never_used_before(jr):
    """ "field_placement" = 1 with chunks of pool size: the first large allocation fills a pool of chunks ("field_pool_pct" of the free memory), every array takes random chunks of it,
    a freed array's chunks go back to the pool and its virtual range is never handed out again (nothing is ever mapped at an address that was mapped before: csrc/fieldpool.hip);
    jrx_field_trim releases what nobody took; jrx_field_list names the arrays."""
    import torch
    from justrelax_jl_amd import _lib, arrays
    h = _lib.Handle(0)
    try:
        h.set_option("field_placement", 1)
        h.set_option("field_chunk_mib", 128)
        h.set_option("field_pool_pct", 1)                      # ~1 % of the free memory: a few dozen chunks of 128 MiB
        h.set_option("scratch_poison", 7)                      # every allocation starts as NaNs: the constructor's fill is what the caller sees
        arrays.use_library_arrays(h)
        shapes = [(257, 130, 67), (1200, 1100), (300, 300, 30), (512, 512, 70)]        # 17.9, 10.6, 21.6 MB: one chunk each; 146.8 MB: two chunks
        rng = np.random.default_rng(3)
        ts, ref = [], []
        for sh in shapes:
            a = rng.standard_normal(sh)
            t = jr.fzeros(sh, "cuda")
            assert float(t.abs().max()) == 0.0
            t.copy_(torch.from_numpy(a).to("cuda"))
            ts.append(t); ref.append(a)
        st = _stats(h)
        pool = st[2]
        assert pool >= 12 and st[3] == pool - 5, st           # the pool was filled once; five chunks are in use
        seen = {t.data_ptr() for t in ts}
        assert len(seen) == 4
        cnt, pl, nb = C.c_int64(), (C.c_void_p * 16)(), (C.c_int64 * 16)()
        h.call("jrx_field_list", C.c_int64(16), pl, nb, C.byref(cnt))
        assert cnt.value == 4 and sorted(pl[i] for i in range(4)) == sorted(seen) and all(nb[i] > 0 for i in range(4))
        for t, a in zip(ts, ref):
            assert np.array_equal(t.cpu().numpy(), a)
        # free and allocate again, several times: chunks come from the pool (none created), addresses are always new, the other arrays keep their contents
        for rep in range(6):
            i = rep % 4
            ts[i] = None
            torch.cuda.synchronize()
            t = jr.fzeros(shapes[i], "cuda", fill=float(rep))
            assert t.data_ptr() not in seen, "a virtual range was handed out twice"
            seen.add(t.data_ptr())
            assert float(t.min()) == float(t.max()) == float(rep)
            ref[i] = rng.standard_normal(shapes[i])
            t.copy_(torch.from_numpy(ref[i]).to("cuda"))
            ts[i] = t
            for u, a in zip(ts, ref):
                assert np.array_equal(u.cpu().numpy(), a)
        st = _stats(h)
        assert st[2] == pool and st[3] == pool - 5, st
        h.call("jrx_field_trim")
        assert _stats(h)[3] == 0
        for u, a in zip(ts, ref):                                  # the arrays in use are untouched by the trim
            assert np.array_equal(u.cpu().numpy(), a)
        # an array made behind the trim, while arrays of the run are live, gets chunks created on the spot -- no second pool (a solve! behind the trim must not spend seconds,
        # and most of the memory, on one)
        created = _stats(h)[2]
        late = jr.fzeros(shapes[0], "cuda", fill=7.0)
        assert _stats(h)[2] == created + 1 and _stats(h)[3] == 0 and float(late.min()) == float(late.max()) == 7.0
        del ts, t, u, late
        torch.cuda.synchronize()
        assert _stats(h)[0] == 0 and _stats(h)[3] == 6             # the freed chunks wait for the next run ...
        h.call("jrx_field_trim")
        assert _stats(h)[3] == 0                                   # ... or for the trim
    finally:
        arrays.use_library_arrays(None)
        h.close()


def test_free_of_a_foreign_pointer_is_an_error(jr):
    import torch
    from justrelax_jl_amd import _lib
    h = _lib.default_handle()
    t = torch.zeros(16, device="cuda", dtype=torch.float64)
    with pytest.raises(_lib.JrxError):
        h.call("jrx_field_free", C.c_void_p(t.data_ptr()))


@pytest.mark.parametrize("placement,chunk,poison", [(1, 2, 0), (1, 128, 7), (2, 64, 0)])
def test_solve_on_library_arrays_gives_the_same_bits(jr, placement, chunk, poison):
    """3D visco-elastic solve, finite dt (fused pipeline: the library's second state set comes from the same pool -- filled with NaNs first in the pool case) on arrays of the
    pool against torch's arrays"""
    from justrelax_jl_amd import _lib, arrays, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    outs = []
    for lib_arrays in (False, True):
        h = _lib.Handle(0)
        try:
            if lib_arrays:
                h.set_option("field_placement", placement)
                h.set_option("field_chunk_mib", chunk)
                h.set_option("field_pool_pct", 1)
                h.set_option("scratch_poison", poison)
                arrays.use_library_arrays(h)
            s = jr.miniapps.random_fields3d((130, 96, 100), seed=11, iterMax=60, nout=20)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs, handle=h)
            outs.append((r, download_stokes(stokes), _stats(h)))
            del stokes, ρg, K, G
        finally:
            arrays.use_library_arrays(None)
            h.close()
    (ra, a, sa), (rb, b, sb) = outs
    assert sa[0] <= 11 and sb[0] > 30, (sa, sb)        # torch's arrays: only the library's own; library arrays: the caller's too
    assert ra.iter == rb.iter and list(ra.err_evo1) == list(rb.err_evo1)
    for k in a:
        m = checks.interior_mask3d(k, a[k].shape)
        assert np.array_equal(a[k][m], b[k][m], equal_nan=True), k
