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


@pytest.mark.parametrize("arena_gib,gap_mib", [(0, 0), (4, 0), (4, 6)])
def test_reroll_keeps_pointers_and_contents_and_the_arena_places_arrays_in_one_range(jr, arena_gib, gap_mib):
    """jrx_tuning_field_reroll gives chunk-backed arrays new physical chunks under the same pointers and carries the contents over (the experiment primitive of round 5); with
    "field_arena_gib" the arrays lie one behind the other in ONE reserved virtual range, "field_va_gap_mib" apart; jrx_field_list names them."""
    import torch
    from justrelax_jl_amd import _lib, arrays
    h = _lib.Handle(0)
    try:
        h.set_option("field_placement", 1)
        h.set_option("field_chunk_mib", 2)
        h.set_option("field_arena_gib", arena_gib)
        h.set_option("field_va_gap_mib", gap_mib)
        arrays.use_library_arrays(h)
        shapes = [(257, 130, 67), (1200, 1100), (300, 300, 30)]        # 17.9, 10.6, 21.6 MB: chunk-backed
        rng = np.random.default_rng(3)
        ts, ref = [], []
        for sh in shapes:
            a = rng.standard_normal(sh)
            t = jr.fzeros(sh, "cuda")
            t.copy_(torch.from_numpy(a).to("cuda"))
            ts.append(t); ref.append(a)
        ptrs0 = [t.data_ptr() for t in ts]
        if arena_gib:
            sz = [-(-int(np.prod(sh)) * 8 // (2 << 20)) * (2 << 20) for sh in shapes]
            assert ptrs0[1] - ptrs0[0] == sz[0] + (gap_mib << 20) and ptrs0[2] - ptrs0[1] == sz[1] + (gap_mib << 20), (ptrs0, sz)
        created0 = _stats(h)[2]
        torch.cuda.synchronize()
        h.call("jrx_tuning_field_reroll", C.c_void_p(ptrs0[1]))          # one array
        h.call("jrx_tuning_field_reroll", C.c_void_p(0))                # all of them
        assert _stats(h)[2] > created0                            # new physical chunks were created for the first re-roll
        assert [t.data_ptr() for t in ts] == ptrs0
        for t, a in zip(ts, ref):
            assert np.array_equal(t.cpu().numpy(), a)
        t2 = ts[0] * 2.0                                          # the re-mapped arrays are ordinary device memory for every later kernel
        assert np.array_equal(t2.cpu().numpy(), ref[0] * 2.0)
        for t, a in zip(ts, ref):                                 # writes behind the re-mapping land where later reads see them (the translation flush of csrc/fieldpool.hip)
            t.copy_(torch.from_numpy(a + 1.0).to("cuda"))
        torch.cuda.synchronize()
        for t, a in zip(ts, ref):
            assert np.array_equal(t.cpu().numpy(), a + 1.0)
        cnt, pl, nb = C.c_int64(), (C.c_void_p * 16)(), (C.c_int64 * 16)()
        h.call("jrx_field_list", C.c_int64(16), pl, nb, C.byref(cnt))
        assert cnt.value == 3 and sorted(pl[i] for i in range(3)) == sorted(ptrs0) and all(nb[i] > 0 for i in range(3))
        with pytest.raises(_lib.JrxError):
            h.call("jrx_tuning_field_reroll", C.c_void_p(ptrs0[0] + 8))
        del ts, t, t2
        torch.cuda.synchronize()
        assert _stats(h)[0] == 0
        # a freed range of the arena is handed out again to an array of the same size
        if arena_gib:
            t = jr.fzeros(shapes[0], "cuda", fill=3.0)
            assert t.data_ptr() == ptrs0[0]
            assert float(t.min()) == float(t.max()) == 3.0
            del t
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


@pytest.mark.parametrize("chunk_mib,pool_pct", [(0, 0), (10, 2)])
def test_placement_search_keeps_addresses_and_a_restored_state_solves_to_the_same_bits(jr, chunk_mib, pool_pct):
    """jrx_stokes3d_tune_placement: draws of new physical chunks under the arrays, the loop body timed on each, the fastest kept.  Pointers stay; the fields are advanced by the probes,
    so the initial state is written back -- and the solve that follows gives the bits of a solve on arrays that were never moved."""
    from justrelax_jl_amd import _lib, arrays, checks, stokes
    from justrelax_jl_amd.miniapps.common import download_stokes, stokes_field_names, upload_stokes, _get
    from justrelax_jl_amd.arrays import from_numpy
    outs = []
    for tune in (False, True):
        h = _lib.Handle(0)
        try:
            h.set_option("field_placement", 1)
            h.set_option("field_chunk_mib", chunk_mib if tune else 2)      # 10 MiB: every array of this problem is ONE chunk of that size -> the draws deal from a pool
            h.set_option("field_pool_pct", pool_pct)
            arrays.use_library_arrays(h)
            s = jr.miniapps.random_fields3d((130, 96, 100), seed=5, iterMax=40, nout=20)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            st, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
            if tune:
                ητ = jr.fzeros(s.ni, st.P.device)
                jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
                ptr0 = [st.P.data_ptr(), st.V.Vx.data_ptr(), st.τ.xy.data_ptr(), ητ.data_ptr()]
                created0 = _stats(h)[2]
                ms, kept = stokes.tune_placement_(st, s.pt, s.grid, s.flow_bcs, ρg, K, G, ητ, s.dt, 3, 4, handle=h)
                assert len(ms) == 5 and all(m > 0 for m in ms) and 0 <= kept <= 3
                import os
                if os.environ.get("JRX_DIAG_FILE"):          # how often a re-mapping had to be flushed and copied a second time (csrc/fieldpool.hip, remap_with)
                    with open(os.environ["JRX_DIAG_FILE"], "a") as fdiag:
                        fdiag.write(f"tune test chunk {chunk_mib} pool {pool_pct}: stat_field_reflushes {h.get_option('stat_field_reflushes')}\n")
                assert [st.P.data_ptr(), st.V.Vx.data_ptr(), st.τ.xy.data_ptr(), ητ.data_ptr()] == ptr0
                assert _stats(h)[2] > created0 and _stats(h)[3] == 0        # draws were made, and the chunks of those that lost went back to the driver
                if pool_pct:
                    assert _stats(h)[2] - created0 > 3 * 40                  # the pool: many more chunks than three draws of the ~40 arrays that took part need
                for name, path in stokes_field_names(3).items():             # the initial state again
                    if name in s.arrays:
                        _get(st, path).copy_(from_numpy(s.arrays[name], st.P.device))
                    else:
                        _get(st, path).zero_()
                del ητ
            r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs, handle=h)
            outs.append((r, download_stokes(st)))
            del st, ρg, K, G
        finally:
            arrays.use_library_arrays(None)
            h.close()
    (ra, a), (rb, b) = outs
    diff = {k: int((a[k] != b[k]).sum()) for k in a if not np.array_equal(a[k][checks.interior_mask3d(k, a[k].shape)], b[k][checks.interior_mask3d(k, a[k].shape)], equal_nan=True)}
    assert ra.iter == rb.iter and list(ra.err_evo1) == list(rb.err_evo1) and not diff, (ra.iter, rb.iter, list(ra.err_evo1), list(rb.err_evo1), diff)
