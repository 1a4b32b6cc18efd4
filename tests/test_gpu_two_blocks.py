"""The N > 1 path of the HIP library with two DIFFERENT blocks, on one device.

Two handles of this process are the two ranks of an ImplicitGlobalGrid decomposition (jrx_comm_init_local: the planes travel by
device-to-device copies ordered by events -- the transport that becomes hipMemcpyPeerAsync between GPUs); each rank's solve! runs on its
own host thread.  Reference being matched: update_halo!(V) inside @hide_communication (src/stokes/Stokes3D.jl:104-121), update_halo!(ητ)
(:57), norm_mpi with the doubly counted overlap (:127-142), ImplicitGlobalGrid's plane selection (src/grid/Utils.jl:26-39); the reference's
own two-rank check is test/test_periodic_boundary_conditions_MPI.jl:9-48 (restated below on the device).

Expectations:
  * uniform material: every block equals the UNDECOMPOSED device run bit for bit on every entry (the duplicated overlap cells stay
    consistent), for the fused pipeline in its three placements of the exchange (beside the kernel = default, behind it, shell tiles) and
    the split sweeps, split along x, y or z;
  * SolVi-style non-uniform viscosity: the decomposed iteration is not the undecomposed one in the reference either -- compute_τ! averages
    η, G to the shear nodes with indices clamped to the LOCAL block (src/MiniKernels.jl:133-147), so the nodes on a block face see one
    cell twice -- hence the expectation is the CPU oracle run block by block with the same plane copies in numpy (tolerance 1e-12 of each
    field's maximum; observed: bit-identical).
"""
import ctypes as C

import numpy as np
import pytest

import _blocks as B

pytestmark = pytest.mark.gpu

STATE = ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "txy", "txz", "tyz")
PIPELINES = {"fused": dict(kernel_variant=3, fused_overlap=0, fused_comm=1), "fused_overlap": dict(kernel_variant=3, fused_overlap=1, fused_comm=1),
             "fused_early": dict(kernel_variant=3, fused_overlap=2, fused_comm=1, comm_bcs_lazy=0),
             "fused_inkernel": dict(kernel_variant=3, fused_overlap=4, fused_comm=1, comm_bcs_lazy=0),      # viscous limit only (dt = Inf); the early exchange otherwise (4: x faces too)
             "fused_inkernel_no_feeder": dict(kernel_variant=3, fused_overlap=4, fused_comm=1, comm_bcs_lazy=0, nbr_feeder=0),     # round 6 A/B: column 0 of a low x face loads the received entries itself
             "fused_inkernel_general4": dict(kernel_variant=3, fused_overlap=3, fused_comm=1, comm_bcs_lazy=0, general_hif=4),     # finite dt: the general form's in-kernel neighbour faces (round 5)
             "fused_inkernel_general3": dict(kernel_variant=3, fused_overlap=3, fused_comm=1, comm_bcs_lazy=0, general_hif=3),
             "fused_early_general0": dict(kernel_variant=3, fused_overlap=3, fused_comm=1, comm_bcs_lazy=0, general_hif=0),        # ... switched off: the early exchange of rounds 3-4
             "fused_inkernel_tall": dict(kernel_variant=3, fused_overlap=4, fused_comm=1, comm_bcs_lazy=0, fused_tile=3),      # ... with the 64 x 8 tile (round 5: the default shape of large blocks)
             "fused_early_tall": dict(kernel_variant=3, fused_overlap=2, fused_comm=1, comm_bcs_lazy=0, fused_tile=3),
             "fused_tall": dict(kernel_variant=3, fused_overlap=0, fused_comm=1, fused_tile=3),
             "fused_early_lazy_bcs": dict(kernel_variant=3, fused_overlap=2, fused_comm=1, comm_bcs_lazy=1),
             "split_sweeps": dict(kernel_variant=3, fused_overlap=0, fused_comm=0)}


def _set(h, **opts):
    for k, v in opts.items():
        h.set_option(k, v)          # options of the public ABI and tuning switches alike


def _get(h, key):
    v = C.c_int64()
    h.call("jrx_get_option", C.c_char_p(key.encode()), C.byref(v))
    return v.value


class TwoBlocks:
    """prod(dims) handles on the current device joined into an in-process group (two ranks in most tests, 2 x 2 x 2 in the last ones)"""

    def __init__(self, n, dims, periods=(0, 0, 0)):
        import torch
        from justrelax_jl_amd import _lib, halo
        self.n, self.dims, self.periods = tuple(n), tuple(dims), tuple(periods)
        self.carts = halo.make_carts(n, dims, periods)
        self.handles = [_lib.Handle(torch.cuda.current_device()) for _ in range(len(self.carts))]
        halo.init_comm_local(self.handles, self.carts)
        self.ng = B.n_global(n, dims, periods)

    def close(self):
        for h in self.handles:
            h.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _global_setup(jr, ng, uniform, iterMax, nout, seed=5, bcs="free_slip", dt=0.25):
    S = jr.miniapps.random_fields3d(ng, seed=seed, iterMax=iterMax, nout=nout, bcs=bcs, dt=dt)
    S.pt.ϵ_rel = S.pt.ϵ_abs = 1e-30
    if uniform:
        for k, v in (("eta", 0.7), ("G", 1.3), ("K", 2.1)):
            S.arrays[k][...] = v
    else:
        # SolVi3D-like: a weak background with a stiff spherical inclusion that straddles the block faces
        x, y, z = (np.linspace(-0.5, 0.5, m) for m in ng)
        r2 = x[:, None, None] ** 2 + y[None, :, None] ** 2 + z[None, None, :] ** 2
        S.arrays["eta"][...] = np.where(r2 < 0.12, 1.0, 1e-2)
    return S


def _solve_blocks(jr, tb, S, pipeline, iters_kw):
    """upload every rank's block of the global setup S and run solve! on all ranks concurrently"""
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd import halo
    from justrelax_jl_amd.miniapps.common import Setup, download_stokes, upload_stokes
    n, ng = tb.n, tb.ng
    g.init_global_grid(*n, dimx=tb.dims[0], dimy=tb.dims[1], dimz=tb.dims[2], periodx=tb.periods[0], periody=tb.periods[1], periodz=tb.periods[2],
                       rank=0, nprocs=len(tb.handles))
    try:
        assert tuple(g.global_grid().n_g(d) for d in range(3)) == ng
        grid = jr.Geometry(n, S.extra["li"])          # geometry_MPI: spacing = li / n_g
        ups = []
        for r, h in enumerate(tb.handles):
            _set(h, **PIPELINES[pipeline])
            loc = Setup(ni=n, arrays={k: B.local_block(v, n, ng, B.coords_of(tb.carts[r])) for k, v in S.arrays.items()})
            ups.append(upload_stokes(loc, jr.AMDGPUBackend))
        fns = [(lambda r=r: jr.solve_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][1], ups[r][2], ups[r][3], S.dt, None, kwargs=iters_kw, handle=tb.handles[r]))
               for r in range(len(tb.handles))]
        res = halo.run_ranks(fns)
        return res, [download_stokes(u[0]) for u in ups]
    finally:
        g.finalize_global_grid()


@pytest.mark.parametrize("pipeline", ["fused", "fused_overlap", "fused_early", "fused_early_lazy_bcs", "split_sweeps", "fused_tall", "fused_early_tall"])
@pytest.mark.parametrize("dims,n", [((2, 1, 1), (70, 13, 12)), ((1, 2, 1), (70, 13, 12)), ((1, 1, 2), (70, 13, 12)),
                                    # 3 x 5 x 5 tiles of the fused kernel per block: every shell box and an interior box
                                    ((2, 1, 1), (130, 14, 40)), ((1, 1, 2), (130, 14, 40))])
def test_two_blocks_equal_the_undecomposed_run_bit_for_bit(jr, dims, n, pipeline):
    """uniform material, 24 iterations with nout = 8 (checks at 8, 16, 24; the last iteration is observed)"""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.checks import interior_mask3d
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        S = _global_setup(jr, tb.ng, True, 23, 8)
        # the undecomposed run on the plain single-rank handle, simplest kernels
        h0 = _lib.default_handle()
        _set(h0, kernel_variant=1)
        try:
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw)
            glob = download_stokes(stokes)
        finally:
            _set(h0, kernel_variant=0)
        res, outs = _solve_blocks(jr, tb, S, pipeline, kw)
        fused_launches = [_get(h, "stat_fused3d") for h in tb.handles]
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    if pipeline == "split_sweeps":
        assert fused_launches == [0, 0]
    else:
        assert min(fused_launches) >= 12, fused_launches       # the fused kernel really ran on both ranks
    for r, out in enumerate(outs):
        co = B.coords_of(tb.carts[r])
        for k in STATE + ("Rx", "Ry", "Rz", "RP", "exx", "exy", "divV", "Ux"):
            want = B.local_block(glob[k], n, tb.ng, co)
            m = interior_mask3d(k, want.shape)
            if k == "Ux":       # U = V dt is taken before flow_bcs! and update_halo! (Stokes3D.jl:116-120): its received planes hold the previous exchange
                m &= B.owned_mask(want.shape, n, tb.carts[r])
            assert np.array_equal(out[k][m], want[m]), (pipeline, dims, r, k, float(np.abs(out[k] - want)[m].max()))
    # norm_mpi: Σ over the ranks of the local interior slices (the 2-cell overlap is counted twice, Stokes3D.jl:127-142) / global counts
    ng = tb.ng
    ss = np.zeros(4)
    for r in range(2):
        co = B.coords_of(tb.carts[r])
        loc = {k: B.local_block(glob[k], n, ng, co) for k in ("Rx", "Ry", "Rz", "RP")}
        ss += [np.sum(loc["Rx"][1:-1, 1:-1, 1:-1] ** 2), np.sum(loc["Ry"][1:-1, 1:-1, 1:-1] ** 2), np.sum(loc["Rz"][1:-1, 1:-1, 1:-1] ** 2), np.sum(loc["RP"] ** 2)]
    cnt = [(ng[0] - 2) * (ng[1] - 1) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 2) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 1) * (ng[2] - 2), ng[0] * ng[1] * ng[2]]
    want_err = max(np.sqrt(ss[q]) / cnt[q] for q in range(4))
    for r in res:
        assert list(r.err_evo2) == [8, 16, 24]
        assert np.isclose(r.err_evo1[-1], want_err, rtol=1e-12), (r.err_evo1[-1], want_err)
    assert list(res[0].err_evo1) == list(res[1].err_evo1)          # every rank holds the same bits (rank-ordered host all-reduce)
    assert getattr(res[0], "norm_∇V")[-1] != getattr(rg, "norm_∇V")[-1]          # ... which are not the undecomposed norm (RP: overlap counted twice)


@pytest.mark.parametrize("pipeline", ["fused", "fused_overlap", "fused_early", "fused_early_lazy_bcs", "fused_inkernel", "fused_inkernel_no_feeder", "split_sweeps"])
@pytest.mark.parametrize("dims,n", [((2, 1, 1), (70, 13, 12)), ((1, 1, 2), (130, 14, 40))])
def test_two_blocks_in_the_viscous_limit_equal_the_undecomposed_general_kernels(jr, dims, n, pipeline):
    """dt = Inf: every rank runs the viscous-limit forms (fused kernel, z-marching sweep, fix-up layers next to received planes), which do not load τ_o, P0, K, G, Q
    -- all non-zero here; the undecomposed run uses the per-node kernels, which load and apply everything"""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.checks import interior_mask3d
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        S = _global_setup(jr, tb.ng, True, 23, 8, seed=11)
        S.dt = np.inf
        h0 = _lib.default_handle()
        _set(h0, kernel_variant=1, viscous_limit=0)
        try:
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw)
            glob = download_stokes(stokes)
        finally:
            _set(h0, kernel_variant=0, viscous_limit=1)
        res, outs = _solve_blocks(jr, tb, S, pipeline, kw)
        assert all(_get(h, "viscous_limit") == 1 for h in tb.handles)
        assert all((_get(h, "stat_fused3d_inkernel") >= 12) == pipeline.startswith("fused_inkernel") for h in tb.handles)
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    for r, out in enumerate(outs):
        co = B.coords_of(tb.carts[r])
        for k in STATE + ("Rx", "Ry", "Rz", "RP", "exx", "exy", "divV"):
            want = B.local_block(glob[k], n, tb.ng, co)
            m = interior_mask3d(k, want.shape)
            assert np.isfinite(want[m]).all() and np.array_equal(out[k][m], want[m]), (pipeline, dims, r, k, float(np.abs(out[k] - want)[m].max()))


@pytest.mark.parametrize("bcs", ["free_slip", "no_slip", "slip_mix", "none"])
@pytest.mark.parametrize("dims,n", [((2, 1, 1), (130, 14, 40)), ((1, 2, 1), (70, 13, 12)), ((1, 2, 1), (70, 40, 12)), ((1, 1, 2), (70, 13, 12)), ((2, 2, 2), (70, 13, 12)), ((2, 2, 1), (130, 14, 40)),
                                    ((1, 2, 2), (130, 14, 40))])
def test_blocks_in_the_viscous_limit_finish_their_neighbour_faces_inside_the_kernel(jr, dims, n, bcs):
    """option fused_overlap = 3 (the default): with dt = Inf the fused kernel's own tiles next to a face with a neighbour wait for the exchange's device-side flag and read
    the received planes themselves -- no flow_bcs! launch, no fix-up launch, the high-face node layers inside the kernel as well.  Two, four and eight blocks (2 x 2 x 2: every
    block has three faces with a neighbour and three physical ones of every kind), single- and multi-tile: every block equals the undecomposed run of the per-node GENERAL
    kernels bit for bit, residuals and strain rates of the observed iterations included."""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.checks import interior_mask3d
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        S = _global_setup(jr, tb.ng, True, 23, 8, seed=21, bcs=bcs, dt=np.inf)
        h0 = _lib.default_handle()
        _set(h0, kernel_variant=1, viscous_limit=0)
        try:
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw)
            glob = download_stokes(stokes)
        finally:
            _set(h0, kernel_variant=0, viscous_limit=1)
        res, outs = _solve_blocks(jr, tb, S, "fused_inkernel", kw)
        launches = [_get(h, "stat_fused3d_inkernel") for h in tb.handles]
    assert min(launches) >= 12, launches
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    assert all(list(r.err_evo1) == list(res[0].err_evo1) for r in res)
    for r, out in enumerate(outs):
        co = B.coords_of(tb.carts[r])
        for k in STATE + ("Rx", "Ry", "Rz", "RP", "exx", "exy", "eyz", "exz", "divV"):
            want = B.local_block(glob[k], n, tb.ng, co)
            m = interior_mask3d(k, want.shape)
            assert np.isfinite(want[m]).all() and np.array_equal(out[k][m], want[m]), (dims, bcs, r, co, k, float(np.abs(out[k] - want)[m].max()))


@pytest.mark.parametrize("pipeline", ["fused_inkernel_tall", "fused_early_tall", "fused_tall"])
@pytest.mark.parametrize("dims,n,bcs", [((2, 1, 1), (130, 30, 40), "free_slip"), ((1, 2, 1), (70, 40, 12), "slip_mix"), ((1, 1, 2), (70, 23, 20), "no_slip"), ((2, 2, 2), (70, 23, 20), "slip_mix")])
def test_blocks_with_the_tall_tile_equal_the_undecomposed_general_kernels(jr, dims, n, bcs, pipeline):
    """the 64 x 8 tile of the viscous-limit fused kernel (tuning switch fused_tile = 3; what large blocks get by default) in the three multi-rank pipelines: every block equals the
    undecomposed run of the per-node general kernels bit for bit"""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.checks import interior_mask3d
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        S = _global_setup(jr, tb.ng, True, 23, 8, seed=5, bcs=bcs, dt=np.inf)
        h0 = _lib.default_handle()
        _set(h0, kernel_variant=1, viscous_limit=0)
        try:
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw)
            glob = download_stokes(stokes)
        finally:
            _set(h0, kernel_variant=0, viscous_limit=1)
        res, outs = _solve_blocks(jr, tb, S, pipeline, kw)
        fused = [_get(h, "stat_fused3d_visc") for h in tb.handles]
    assert min(fused) >= 12, fused
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    for r, out in enumerate(outs):
        co = B.coords_of(tb.carts[r])
        for k in STATE + ("Rx", "Ry", "Rz", "RP", "exx", "exy", "eyz", "exz", "divV"):
            want = B.local_block(glob[k], n, tb.ng, co)
            m = interior_mask3d(k, want.shape)
            assert np.isfinite(want[m]).all() and np.array_equal(out[k][m], want[m]), (dims, bcs, pipeline, r, co, k, float(np.abs(out[k] - want)[m].max()))


@pytest.mark.parametrize("pipeline", ["fused_inkernel_general4", "fused_inkernel_general3", "fused_early_general0"])
@pytest.mark.parametrize("dims,n,bcs,zero", [((2, 1, 1), (130, 14, 40), "free_slip", ""), ((1, 2, 1), (97, 40, 12), "slip_mix", "xy"), ((1, 1, 2), (70, 13, 12), "no_slip", ""), ((2, 2, 2), (130, 13, 20), "slip_mix", "xyz"),
                                             ((2, 2, 1), (130, 14, 40), "none", "")])
def test_blocks_with_finite_dt_finish_their_neighbour_faces_inside_the_general_kernel(jr, dims, n, bcs, zero, pipeline):
    """the GENERAL form of the fused kernel (finite dt: old stresses, P0, K, G, Q all in play) with its high-face node layers inside and, with neighbours, its boundary
    tiles reading the received planes -- no flow_bcs! launch, no fix-up -- built for four or three waves per SIMD (tuning switch general_hif).  Two, four and eight blocks equal the
    undecomposed run of the per-node kernels bit for bit, residuals and strain rates of the observed iterations included; with general_hif = 0 the early exchange of rounds 3-4 runs."""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.checks import interior_mask3d
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        S = _global_setup(jr, tb.ng, True, 23, 8, seed=9, bcs=bcs, dt=0.25)
        for c in zero:
            S.arrays["f" + c][...] = 0.0
        h0 = _lib.default_handle()
        _set(h0, kernel_variant=1)
        try:
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw)
            glob = download_stokes(stokes)
        finally:
            _set(h0, kernel_variant=0)
        res, outs = _solve_blocks(jr, tb, S, pipeline, kw)
        inkernel = [_get(h, "stat_fused3d_inkernel") for h in tb.handles]
        ghif = [_get(h, "stat_fused3d_general_hif") for h in tb.handles]
    if pipeline == "fused_early_general0" or n[0] <= 90:          # (nx = 63 .. 90 runs 32 x 8 tiles, for which the one-launch general form is not instantiated: early exchange)
        assert max(inkernel) == 0 and max(ghif) == 0, (inkernel, ghif)
    else:
        assert min(inkernel) >= 12 and min(ghif) >= 12, (inkernel, ghif)
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    assert all(list(r.err_evo1) == list(res[0].err_evo1) for r in res)
    for r, out in enumerate(outs):
        co = B.coords_of(tb.carts[r])
        for k in STATE + ("Rx", "Ry", "Rz", "RP", "exx", "exy", "eyz", "exz", "divV"):
            want = B.local_block(glob[k], n, tb.ng, co)
            m = interior_mask3d(k, want.shape)
            assert np.isfinite(want[m]).all() and np.array_equal(out[k][m], want[m]), (dims, bcs, pipeline, r, co, k, float(np.abs(out[k] - want)[m].max()))


@pytest.mark.parametrize("zero,nof", [("xy", 1), ("xyz", 2)])
@pytest.mark.parametrize("dims,n,bcs", [((2, 1, 1), (130, 14, 40), "free_slip"), ((1, 1, 2), (70, 13, 12), "slip_mix"), ((2, 2, 2), (70, 13, 12), "no_slip")])
def test_blocks_whose_body_forces_are_zero_do_not_load_them_and_keep_the_bits(jr, dims, n, bcs, zero, nof):
    """ρg_x = ρg_y = +0.0 (gravity along z: every 3D model of the reference) or all three (SolVi3D.jl:102): the one-launch viscous-limit kernel with the neighbour faces inside
    does not load those arrays (k_fused3d, NOF) -- every block still equals the undecomposed run of the per-node general kernels, which subtract the zeros, bit for bit."""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.checks import interior_mask3d
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        S = _global_setup(jr, tb.ng, True, 23, 8, seed=22, bcs=bcs, dt=np.inf)
        for c in zero:
            S.arrays["f" + c][...] = 0.0
        h0 = _lib.default_handle()
        _set(h0, kernel_variant=1, viscous_limit=0)
        try:
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw)
            glob = download_stokes(stokes)
        finally:
            _set(h0, kernel_variant=0, viscous_limit=1)
        res, outs = _solve_blocks(jr, tb, S, "fused_inkernel", kw)
        launches = [(_get(h, "stat_fused3d_inkernel"), _get(h, "stat_fused3d_nof1"), _get(h, "stat_fused3d_nof2")) for h in tb.handles]
    for l in launches:
        assert l[0] >= 12 and l[nof] == l[0] and l[3 - nof] == 0, launches
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    for r, out in enumerate(outs):
        co = B.coords_of(tb.carts[r])
        for k in STATE + ("Rx", "Ry", "Rz", "RP"):
            want = B.local_block(glob[k], n, tb.ng, co)
            m = interior_mask3d(k, want.shape)
            assert np.isfinite(want[m]).all() and np.array_equal(out[k][m], want[m]), (dims, bcs, r, co, k, float(np.abs(out[k] - want)[m].max()))


@pytest.mark.parametrize("pipeline", ["fused", "fused_overlap", "fused_early", "split_sweeps"])
@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1), (1, 1, 2)])
def test_two_blocks_with_an_inclusion_match_the_oracle_block_by_block(jr, oracle, dims, pipeline):
    """SolVi-style non-uniform η (with its ητ halo): device blocks == oracle blocks + numpy plane copies"""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.checks import interior_mask3d
    orc = oracle
    L = _lib.load()
    n = (70, 13, 12)
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        S = _global_setup(jr, tb.ng, False, 23, 8, seed=9)
        res, outs = _solve_blocks(jr, tb, S, pipeline, kw)
    ng = tb.ng
    b = S.flow_bcs
    pl = orc.params3d(n, S.grid._di["center"], S.dt, dict(r=S.pt.r, theta_dtau=S.pt.θ_dτ, eta_dtau=S.pt.ηdτ, eps_rel=1e-30, eps_abs=1e-30),
                      iterMax=23, nout=8, free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, ni_g=ng)
    loc = [{k: B.local_block(v, n, ng, B.coords_of(tb.carts[r])) for k, v in S.arrays.items()} for r in range(2)]
    et = [orc.compute_maxloc(l["eta"]) for l in loc]
    B.exchange([[e] for e in et], n, tb.carts, L)
    errs = []
    for it in range(1, 25):
        for r in range(2):
            orc.stokes3d_iteration(loc[r], et[r], pl)
        B.exchange([[l["Vx"], l["Vy"], l["Vz"]] for l in loc], n, tb.carts, L)
        if it % 8 == 0:
            ss = sum(orc.residual_sumsq3d(l, pl) for l in loc)
            cnt = [(ng[0] - 2) * (ng[1] - 1) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 2) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 1) * (ng[2] - 2), ng[0] * ng[1] * ng[2]]
            errs.append(max(np.sqrt(ss[q]) / cnt[q] for q in range(4)))
    for r in range(2):
        assert res[r].iter == 24
        assert np.allclose(res[r].err_evo1, errs, rtol=1e-11), (res[r].err_evo1, errs)
        for k in STATE:
            m = interior_mask3d(k, loc[r][k].shape)
            scale = np.abs(loc[r][k]).max()
            d = np.abs(outs[r][k] - loc[r][k])[m].max()
            assert d <= 1e-12 * scale, (pipeline, dims, r, k, d, scale)
    # the inclusion makes the decomposed iteration differ from the undecomposed one (clamped η averages on the block face): make sure the
    # test would notice, i.e. that the blocks really differ from slices of an undecomposed oracle run
    glob = {k: v.copy(order="F") for k, v in S.arrays.items()}
    from justrelax_jl_amd import checks
    pg = checks.oracle_params3d(orc, S)
    etg = orc.compute_maxloc(glob["eta"])
    for it in range(24):
        orc.stokes3d_iteration(glob, etg, pg)
    d = max(np.abs(B.local_block(glob["Vx"], n, ng, B.coords_of(tb.carts[r])) - loc[r]["Vx"]).max() for r in range(2))
    assert d > 1e-9


def test_reference_two_rank_periodic_check_on_the_device(jr):
    """test/test_periodic_boundary_conditions_MPI.jl:9-48 in 3D on two device blocks: dims = (2, 1, 1), periodic in x; every field holds
    coords[1] + 1; after the BCs and update_halo! the first and the last x plane hold the OTHER rank's value."""
    import torch
    from justrelax_jl_amd import halo
    from justrelax_jl_amd.arrays import from_numpy, to_numpy
    n = (8, 6, 5)
    dev = torch.device("cuda", torch.cuda.current_device())
    with TwoBlocks(n, (2, 1, 1), periods=(1, 0, 0)) as tb:
        assert tb.ng[0] == 2 * (n[0] - 2)
        shapes = [(n[0] + 1, n[1] + 2, n[2] + 2), (n[0] + 2, n[1] + 1, n[2] + 2), (n[0] + 2, n[1] + 2, n[2] + 1), (n[0] + 2, n[1] + 2, n[2] + 2)]
        fields = [[from_numpy(np.full(s, float(r + 1), order="F"), dev) for s in shapes] for r in range(2)]
        torch.cuda.synchronize()
        halo.run_ranks([(lambda r=r: halo.update_halo_(*fields[r], ni=n, handle=tb.handles[r])) for r in range(2)])
        torch.cuda.synchronize()
        for r in range(2):
            other = float(2 - r)
            for f in fields[r]:
                A = to_numpy(f)
                assert (A[0] == other).all() and (A[-1] == other).all()
                assert (A[1:-1] == float(r + 1)).all()


def test_local_group_reports_an_absent_rank_instead_of_hanging(jr):
    """a rank whose neighbour never calls: JRX_ERR_RCCL after the time-out, and the group stays failed"""
    import torch
    from justrelax_jl_amd import _lib, halo
    from justrelax_jl_amd.arrays import from_numpy
    n = (8, 6, 5)
    dev = torch.device("cuda", torch.cuda.current_device())
    with TwoBlocks(n, (2, 1, 1)) as tb:
        tb.handles[0].call("jrx_tuning_set", C.c_char_p(b"comm_timeout_ms"), C.c_int64(300))
        A = from_numpy(np.zeros((n[0] + 1, n[1] + 2, n[2] + 2), order="F"), dev)
        with pytest.raises(_lib.JrxError) as e:
            halo.update_halo_(A, ni=n, handle=tb.handles[0])
        assert e.value.status == 3 and "timed out" in str(e.value)


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("hide", [2, 1, 0])
@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1), (1, 1, 2)])
def test_vep3d_two_blocks_equal_the_undecomposed_run(jr, dims, hide, fuse):
    """jrx_stokes3d_vep_solve on two blocks: update_halo!(ητ), update_halo!(τ.yz / τ.xz / τ.xy) and update_halo!(V) every iteration, norms over the
    ranks (Stokes3D.jl:515,578-580,582-597,607-612).  Two phases of equal viscosity but different G and cohesion, pre-stressed to yield: with a
    uniform η the clamped centre-to-edge averages on a block face only touch the edge nodes of the outermost plane, which the τ exchange replaces,
    so every block must equal the undecomposed device run bit for bit on the state arrays.  hide = 2 (default): the three exchanges run on the halo
    stream beside independent kernels (@hide_communication of Stokes3D.jl:582-597 and more); hide = 1: ητ and the edge stresses only; hide = 0: in order on the compute stream.
    fuse = 1 (default): the blocks run pre, viscosity relaxation and centre pass as one kernel (k_vep3_prec<ML = false>, ητ from the exchanged array); 0: the three kernels."""
    import test_gpu_vep3d as tv
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd import halo
    from justrelax_jl_amd.checks import interior_mask3d
    from justrelax_jl_amd.miniapps.common import Setup
    n = (40, 13, 12)
    kw = dict(iterMax=23, nout=8, verbose=False, viscosity_cutoff=(-np.inf, np.inf))
    with TwoBlocks(n, dims) as tb:
        ng = tb.ng
        S = jr.miniapps.shearband3d(ng, iterMax=23, nout=8)
        S.pt.ϵ_rel = S.pt.ϵ_abs = 1e-30
        S.extra["phases"][1]["eta"] = S.extra["phases"][0]["eta"]
        S.arrays["eta"][...] = S.extra["phases"][0]["eta"]
        rng = np.random.default_rng(3)
        for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):     # pre-stress close to yield so that plasticity is active
            S.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=S.arrays["to" + c].shape)
            S.arrays["t" + c][...] = S.arrays["to" + c]
        stokes, pr, ρg = tv._upload(jr, S)
        rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, pr, S.extra["phases"], None, S.dt, None, kwargs=kw)
        glob = tv._download(jr, stokes)
        g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=dims[2], rank=0, nprocs=2)
        try:
            grid = jr.Geometry(n, S.extra["li"], origin=(0.0, 0.0, 0.0))
            ups = []
            for r in range(2):
                tb.handles[r].set_option("vep3_hide_comm", hide)
                tb.handles[r].set_option("vep3_fuse_pc", fuse)
                f0 = tb.handles[r].get_option("stat_vep3_fused")
                loc = Setup(ni=n, arrays={k: B.local_block(v, n, ng, B.coords_of(tb.carts[r])) for k, v in S.arrays.items()})
                ups.append(tv._upload(jr, loc))
            res = halo.run_ranks([(lambda r=r: jr.solve_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][2], ups[r][1], S.extra["phases"], None, S.dt, None, kwargs=kw,
                                                         handle=tb.handles[r])) for r in range(2)])
            outs = [tv._download(jr, u[0]) for u in ups]
            assert (tb.handles[1].get_option("stat_vep3_fused") > f0) == bool(fuse)
        finally:
            g.finalize_global_grid()
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    assert (glob["eplxx"] != 0).any() and (glob["eplxx"] == 0).any()
    assert list(res[0].err_evo1) == list(res[1].err_evo1) and len(res[0].err_evo1) == 3
    for r in range(2):
        co = B.coords_of(tb.carts[r])
        for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "tyz", "txz", "txy", "tyz_c", "txz_c", "txy_c", "tII", "eta_vep", "eta", "exx", "exz", "eplzz",
                  "Rx", "Rz", "RP", "toxx", "toxz", "toxy_c", "EII_pl"):
            want = B.local_block(glob[k], n, ng, co)
            m = interior_mask3d(k, want.shape)
            if k == "EII_pl":     # accumulate_tensor! gathers the edge plastic strains, which nobody exchanges: the cell layer on a block face is local
                m &= B.owned_mask(want.shape, n, tb.carts[r])
            assert np.array_equal(outs[r][k][m], want[m]), (dims, r, k, float(np.abs(outs[r][k] - want)[m].max()))


@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1), (1, 1, 2)])
def test_thermal3d_two_blocks_equal_the_undecomposed_run(jr, dims):
    """jrx_heatdiffusion_PT3d on two blocks: update_halo!(thermal.T) every iteration (DiffusionPT_solver.jl:110), uniform K and ρCp"""
    import torch
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd import halo
    from justrelax_jl_amd.arrays import from_numpy
    n = (24, 13, 12)
    kw = dict(iterMax=40, nout=20, verbose=False)
    dev = torch.device("cuda", torch.cuda.current_device())

    def run(ni, arrays, grid, handle, li, di):
        thermal = jr.ThermalArrays(jr.AMDGPUBackend, ni)
        for name in ("T", "H"):
            getattr(thermal, name).copy_(from_numpy(arrays[name], dev))
        K, ρCp = from_numpy(arrays["K"], dev), from_numpy(arrays["rhoCp"], dev)
        pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, S.dt, di, li, CFL=S.pt["CFL"], ϵ=1e-30)
        return thermal, (lambda: jr.heatdiffusion_PT_(thermal, pt, S.flow_bcs, K, ρCp, S.dt, grid, kwargs=kw, handle=handle))

    with TwoBlocks(n, dims) as tb:
        ng = tb.ng
        S = jr.miniapps.diffusion3d(ng, iterMax=40, nout=20)
        rng = np.random.default_rng(8)
        S.arrays["T"][...] += rng.uniform(-50.0, 50.0, size=S.arrays["T"].shape)
        S.arrays["K"][...] = S.arrays["K"].flat[0]
        S.arrays["rhoCp"][...] = S.arrays["rhoCp"].flat[0]
        tg, fg = run(ng, S.arrays, S.grid, None, S.extra["li"], S.extra["di"])
        rg = fg()
        Tg = jr.to_numpy(tg.T)
        g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=dims[2], rank=0, nprocs=2)
        try:
            grid = jr.Geometry(n, S.extra["li"])
            runs = []
            for r in range(2):
                co = B.coords_of(tb.carts[r])
                loc = {k: B.local_block(S.arrays[k], n, ng, co) for k in ("T", "H", "K", "rhoCp")}
                runs.append(run(n, loc, grid, tb.handles[r], S.extra["li"], S.extra["di"]))
            res = halo.run_ranks([f for _, f in runs])
            Ts = [jr.to_numpy(t.T) for t, _ in runs]
            Rs = [jr.to_numpy(t.ResT) for t, _ in runs]
        finally:
            g.finalize_global_grid()
    assert list(rg.iter_count) == [20, 40] and all(list(r.iter_count) == [20, 40] for r in res)
    for r in range(2):
        want = B.local_block(Tg, n, ng, B.coords_of(tb.carts[r]))
        inner = (slice(1, -1),) * 3
        assert np.array_equal(Ts[r][inner], want[inner]), (dims, r, float(np.abs(Ts[r] - want)[inner].max()))
    # the reference reports each rank's LOCAL norm(ResT) * _sq_len_RT (DiffusionPT_solver.jl:131, no MPI reduction); the ranks leave together
    for r in range(2):
        assert np.isclose(res[r].norm_ResT[-1], np.linalg.norm(Rs[r].ravel()) / np.sqrt(Rs[r].size), rtol=1e-12)
    assert res[0].norm_ResT[-1] != res[1].norm_ResT[-1]


@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1)])
def test_stokes2d_two_blocks_equal_the_undecomposed_run(jr, dims):
    """jrx_stokes2d_solve on two blocks (update_halo!(ητ), update_halo!(V) every iteration, norm_mpi; Stokes2D.jl:209,268,277-300), uniform material:
    every block equals the undecomposed device run bit for bit"""
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd import halo
    from justrelax_jl_amd.miniapps.common import Setup, download_stokes, upload_stokes
    n = (40, 23, 1)
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        ng = tb.ng
        S = jr.miniapps.random_fields2d(ng[:2], seed=6, iterMax=23, nout=8)
        S.pt.ϵ_rel = S.pt.ϵ_abs = 1e-30
        for k, v in (("eta", 0.7), ("G", 1.3), ("K", 2.1)):
            S.arrays[k][...] = v
        stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
        rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, G, K, S.dt, None, kwargs=kw)
        glob = download_stokes(stokes)
        g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=1, rank=0, nprocs=2)
        try:
            grid = jr.Geometry(n[:2], S.extra["li"])
            ups = []
            for r in range(2):
                loc = Setup(ni=n[:2], arrays={k: B.local_block(v, n, ng, B.coords_of(tb.carts[r]), nd=2) for k, v in S.arrays.items()})
                ups.append(upload_stokes(loc, jr.AMDGPUBackend))
            res = halo.run_ranks([(lambda r=r: jr.solve_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][1], ups[r][3], ups[r][2], S.dt, None, kwargs=kw,
                                                         handle=tb.handles[r])) for r in range(2)])
            outs = [download_stokes(u[0]) for u in ups]
        finally:
            g.finalize_global_grid()
    assert rg.iter == 24 and all(r.iter == 24 for r in res) and list(res[0].err_evo1) == list(res[1].err_evo1)
    for r in range(2):
        for k in ("P", "Vx", "Vy", "txx", "tyy", "txy", "Rx", "Ry", "RP", "exx", "exy"):
            want = B.local_block(glob[k], n, ng, B.coords_of(tb.carts[r]), nd=2)
            m = np.ones(want.shape, dtype=bool)
            if k in ("Vx", "Vy"):       # corner ghosts are never read
                gd = 1 if k == "Vx" else 0
                cm = np.zeros(want.shape, dtype=int)
                idx = [slice(None)] * 2
                for e in (0, want.shape[gd] - 1):
                    idx[gd] = e
                    cm[tuple(idx)] += 1
                od = 1 - gd
                idx = [slice(None)] * 2
                for e in (0, want.shape[od] - 1):
                    idx[od] = e
                    cm[tuple(idx)] += 1
                m = cm < 2
            assert np.array_equal(outs[r][k][m], want[m]), (dims, r, k, float(np.abs(outs[r][k] - want)[m].max()))


@pytest.mark.parametrize("pipeline", ["fused", "fused_early", "split_sweeps"])
@pytest.mark.parametrize("dims,n", [((2, 2, 2), (70, 13, 12)), ((2, 2, 1), (70, 13, 12)), ((1, 2, 2), (130, 14, 40))])
def test_eight_blocks_2x2x2_equal_the_undecomposed_run(jr, dims, n, pipeline):
    """BASELINE configs[3] in miniature: the 2 x 2 x 2 block decomposition of north_star (eight ranks = eight handles of this process on ONE device,
    each on its own host thread), where every block has three faces with a neighbour and the edge / corner values of the ghost planes only arrive through the
    x -> y -> z order of update_halo! (a two-rank run cannot see a mistake there).  Uniform material: every block equals the undecomposed device run bit for bit."""
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    from justrelax_jl_amd.checks import interior_mask3d
    kw = dict(iterMax=23, nout=8, verbose=False)
    with TwoBlocks(n, dims) as tb:
        nr = len(tb.handles)
        assert nr == int(np.prod(dims))
        S = _global_setup(jr, tb.ng, True, 23, 8, seed=12)
        h0 = _lib.default_handle()
        _set(h0, kernel_variant=1)
        try:
            stokes, ρg, K, G = upload_stokes(S, jr.AMDGPUBackend)
            rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, K, G, S.dt, None, kwargs=kw)
            glob = download_stokes(stokes)
        finally:
            _set(h0, kernel_variant=0)
        res, outs = _solve_blocks(jr, tb, S, pipeline, kw)
        faces = [sum(1 for d in range(3) for sd in range(2) if tb.carts[r].neighbor[d][sd] >= 0) for r in range(nr)]
    assert all(f == sum(1 for d in dims if d > 1) for f in faces)          # every block: one neighbour per split dimension
    assert rg.iter == 24 and all(r.iter == 24 for r in res)
    assert all(list(r.err_evo1) == list(res[0].err_evo1) for r in res)
    for r, out in enumerate(outs):
        co = B.coords_of(tb.carts[r])
        for k in STATE + ("Rx", "Ry", "Rz", "RP"):
            want = B.local_block(glob[k], n, tb.ng, co)
            m = interior_mask3d(k, want.shape)
            assert np.array_equal(out[k][m], want[m]), (pipeline, dims, r, co, k, float(np.abs(out[k] - want)[m].max()))


@pytest.mark.parametrize("dims", [(2, 1, 1), (1, 2, 1)])
def test_vep2d_two_blocks_equal_the_undecomposed_run(jr, dims):
    """jrx_stokes2d_vep_solve on two blocks: update_halo!(ητ), update_halo!(τ.xy), update_halo!(V) every iteration (Stokes2D.jl:655,757,784), norm_mpi.
    Two phases of equal viscosity (different G and cohesion), pre-stressed to yield: every block equals the undecomposed device run bit for bit on the state arrays."""
    import test_gpu_vep2d as tv
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd import halo
    from justrelax_jl_amd.miniapps.common import Setup
    n = (34, 66, 1) if dims[0] == 2 else (66, 34, 1)            # a square 66 x 66 global grid either way
    kw = dict(iterMax=29, nout=10, iterMin=5, verbose=False)
    with TwoBlocks(n, dims) as tb:
        ng = tb.ng
        assert ng[:2] == (66, 66)
        S = jr.miniapps.shearband2d(66, iterMax=29, nout=10)
        S.pt.ϵ_rel = S.pt.ϵ_abs = 1e-30
        S.extra["phases"][1]["eta"] = S.extra["phases"][0]["eta"]
        S.arrays["eta"][...] = S.extra["phases"][0]["eta"]
        rng = np.random.default_rng(13)
        for c in ("xx", "yy", "xy", "xy_c"):
            S.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=S.arrays["to" + c].shape)
            S.arrays["t" + c][...] = S.arrays["to" + c]
        stokes, pr, ρg = tv._upload(jr, S)
        rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, ρg, pr, S.extra["phases"], None, S.dt, None, kwargs=kw)
        glob = tv._download(jr, stokes)
        g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=1, rank=0, nprocs=2)
        try:
            grid = jr.Geometry(n[:2], (1.0, 1.0))
            ups = []
            for r in range(2):
                loc = Setup(ni=n[:2], arrays={k: B.local_block(v, n, ng, B.coords_of(tb.carts[r]), nd=2) for k, v in S.arrays.items()})
                ups.append(tv._upload(jr, loc))
            res = halo.run_ranks([(lambda r=r: jr.solve_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][2], ups[r][1], S.extra["phases"], None, S.dt, None, kwargs=kw,
                                                         handle=tb.handles[r])) for r in range(2)])
            outs = [tv._download(jr, u[0]) for u in ups]
        finally:
            g.finalize_global_grid()
    assert rg.iter == res[0].iter == res[1].iter and list(res[0].err_evo1) == list(res[1].err_evo1)
    assert (glob["eplxx"] != 0).any()
    for r in range(2):
        for k in ("P", "txx", "tyy", "txy", "tII", "eta_vep", "exx", "Rx", "Ry", "RP"):
            want = B.local_block(glob[k], n, ng, B.coords_of(tb.carts[r]), nd=2)
            d = np.abs(outs[r][k] - want)
            bad = np.argwhere(d > 0)
            assert np.array_equal(outs[r][k], want), (dims, r, k, float(d.max()), want.shape, bad[:6].tolist(), bad[-3:].tolist(), len(bad))
        for k in ("Vx", "Vy"):
            want = B.local_block(glob[k], n, ng, B.coords_of(tb.carts[r]), nd=2)
            inner = (slice(1, -1), slice(1, -1))
            assert np.array_equal(outs[r][k][inner], want[inner]), (dims, r, k)


def test_two_blocks_on_pool_placed_library_arrays_keep_the_bits(jr):
    """coupled blocks whose arrays (the callers' and the libraries' own second state sets, filled with NaNs first) come from the placement pool of their handles
    ("field_placement" = 1, csrc/fieldpool.hip) solve to the bits of the same blocks on torch's arrays"""
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd import arrays, halo
    from justrelax_jl_amd.miniapps.common import Setup, download_stokes, upload_stokes
    dims, n = (2, 1, 1), (130, 96, 100)
    kw = dict(iterMax=30, nout=10, verbose=False)
    outs = []
    for pooled in (False, True):
        with TwoBlocks(n, dims) as tb:
            S = _global_setup(jr, tb.ng, False, 30, 10)
            g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=dims[2], rank=0, nprocs=2)
            try:
                grid = jr.Geometry(n, S.extra["li"])
                ups = []
                for r, h in enumerate(tb.handles):
                    _set(h, **PIPELINES["fused_early"])
                    if pooled:
                        h.set_option("field_placement", 1)
                        h.set_option("field_chunk_mib", 128)
                        h.set_option("field_pool_pct", 1)
                        h.set_option("scratch_poison", 7)
                        arrays.use_library_arrays(h)
                    loc = Setup(ni=n, arrays={k: B.local_block(v, n, tb.ng, B.coords_of(tb.carts[r])) for k, v in S.arrays.items()})
                    ups.append(upload_stokes(loc, jr.AMDGPUBackend))
                    arrays.use_library_arrays(None)
                solve = lambda r, k: jr.solve_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][1], ups[r][2], ups[r][3], S.dt, None, kwargs=dict(kw, iterMax=k), handle=tb.handles[r])
                res = halo.run_ranks([(lambda r=r: solve(r, 30)) for r in range(2)])
                outs.append((res, [download_stokes(u[0]) for u in ups]))
                del ups
            finally:
                arrays.use_library_arrays(None)
                g.finalize_global_grid()
    (ra, a), (rb, b) = outs
    for r in range(2):
        assert ra[r].iter == rb[r].iter and list(ra[r].err_evo1) == list(rb[r].err_evo1)
        for k in STATE:
            assert np.array_equal(a[r][k], b[r][k], equal_nan=True), (r, k)


def test_compute_dt_takes_the_maximum_over_the_ranks(jr):
    """compute_dt(stokes, di, dt_diff, igg) (Utils.jl:492-519): maximum_mpi over the group -- both ranks get the dt of the faster block"""
    import torch
    from justrelax_jl_amd import halo
    n = (12, 10, 9)
    with TwoBlocks(n, (2, 1, 1)) as tb:
        sts = []
        for r in range(2):
            st = jr.StokesArrays(jr.AMDGPUBackend, n)
            st.V.Vx.fill_(0.5 * (r + 1)); st.V.Vy.fill_(0.1); st.V.Vz.fill_(-0.2 * (r + 1))
            sts.append(st)
        torch.cuda.synchronize()
        di = (0.1, 0.2, 0.3)
        dts = halo.run_ranks([(lambda r=r: jr.compute_dt_(sts[r], di, float("inf"), None, handle=tb.handles[r])) for r in range(2)])
        alone = jr.compute_dt_(sts[0], di, float("inf"), None)          # the plain handle: no communicator
    want = 0.9 * min(0.1 / 1.0, 0.2 / 0.1, 0.3 / 0.4)
    assert dts[0] == dts[1] == pytest.approx(want, rel=1e-15)
    assert alone == pytest.approx(0.9 * min(0.1 / 0.5, 0.2 / 0.1, 0.3 / 0.2), rel=1e-15) and alone != dts[0]


def test_comm_init_local_rejects_bad_arguments(jr):
    """jrx_comm_init_local: carts must be rank r of n in order, handles distinct; a refused call leaves no communicator behind"""
    import torch
    from justrelax_jl_amd import _lib, halo
    hs = [_lib.Handle(torch.cuda.current_device()) for _ in range(2)]
    try:
        carts = halo.make_carts((8, 6, 5), (2, 1, 1))
        swapped = (_lib.Cart * 2)(carts[1], carts[0])
        with pytest.raises(_lib.JrxError) as e:
            halo.init_comm_local(hs, swapped)
        assert e.value.status == 4 and "carts[0] is rank 1" in str(e.value)
        with pytest.raises(_lib.JrxError) as e:
            halo.init_comm_local([hs[0], hs[0]], carts)
        assert e.value.status == 4 and "appears twice" in str(e.value)
        cnt = C.c_int32(-1)
        hs[0].call("jrx_comm_count", C.byref(cnt))
        assert cnt.value == 0
        halo.init_comm_local(hs, carts)
        hs[1].call("jrx_comm_count", C.byref(cnt))
        assert cnt.value == 2
        # leaving the group breaks it for the rank that stays: its next exchange reports the failure at once instead of waiting for the time-out
        hs[1].call("jrx_comm_destroy")
        from justrelax_jl_amd.arrays import from_numpy
        A = from_numpy(np.zeros((9, 8, 7), order="F"), torch.device("cuda", torch.cuda.current_device()))
        with pytest.raises(_lib.JrxError) as e:
            halo.update_halo_(A, ni=(8, 6, 5), handle=hs[0])
        assert e.value.status == 3 and "failed or left" in str(e.value)
    finally:
        for h in hs:
            h.close()
