"""jrx_field_alloc / jrx_field_free: the library hands out the state arrays (the backend owns the array constructor in the reference:
src/ext/AMDGPU/3D.jl:46-48 StokesArrays(::Type{AMDGPUBackend}, ni) -> @zeros(ni...)).  Whatever "field_placement" selects -- hipMalloc,
physical chunks mapped in shuffled order, physically contiguous memory -- an array is plain device memory: a solve on library arrays gives
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
        # a freed range can be handed out again (chunks come back from the spare list: no new ones)
        created = _stats(h)[2]
        t = jr.fzeros(shapes[0], "cuda")
        assert _stats(h)[2] == created
        assert p0 != 0 and t.data_ptr() != 0
        del t
        h.call("jrx_field_trim")
        assert _stats(h)[3] == 0
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


@pytest.mark.parametrize("placement", [1, 2])
def test_solve_on_library_arrays_gives_the_same_bits(jr, placement):
    """3D visco-elastic solve (fused pipeline: the library's second state set comes from the same pool) on arrays of the pool against torch's arrays"""
    from justrelax_jl_amd import _lib, arrays, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    outs = []
    for lib_arrays in (False, True):
        h = _lib.Handle(0)
        try:
            if lib_arrays:
                h.set_option("field_placement", placement)
                h.set_option("field_chunk_mib", 2)
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
