"""GPU tests of update_halo! (ImplicitGlobalGrid semantics) on one device.

A 1-GPU box has no neighbour rank, so the exchange is exercised through periodic dimensions held by the rank
itself (IGG copies locally in that case): once through the library's local-copy path and once, with the test hook
option halo_self_rccl = 1, through a one-rank RCCL communicator -- the same pack kernel -> grouped ncclSend/ncclRecv ->
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
    h = _lib.Handle(torch.cuda.current_device())
    try:
        halo.init_comm(h, self_rccl=through_rccl)
        dev = [from_numpy(a, torch.device('cuda', torch.cuda.current_device())) for a in host]
        halo.update_halo_(*dev, ni=n, handle=h)
        torch.cuda.synchronize()
        got = [to_numpy(d) for d in dev]
    finally:
        h.close()
        g.finalize_global_grid()
    exp = _expected(host, n, periods, L)
    for a, b, s in zip(got, exp, shapes):
        assert np.array_equal(a, b), f"halo mismatch for array of shape {s}, periods {periods}, rccl={through_rccl}"
        # interior untouched
    assert any(not np.array_equal(a, b) for a, b in zip(got, host))


@pytest.mark.parametrize("through_rccl,variant,periods,n", [(False, 2, (1, 0, 1), (70, 13, 12)), (True, 2, (1, 0, 1), (70, 13, 12)),
                                                            (False, 3, (1, 0, 1), (70, 13, 12)), (True, 3, (1, 1, 1), (70, 13, 12)),
                                                            (False, 3, (0, 1, 0), (70, 13, 12)),
                                                            # 3 x 5 x 3 tiles of the fused kernel: all six shell boxes + an interior box
                                                            (True, 3, (1, 1, 1), (130, 14, 40)), (False, 0, (1, 0, 1), (130, 14, 40)),
                                                            # 13 = variant 3 with option fused_overlap (shell on the halo stream)
                                                            (True, 13, (1, 1, 1), (130, 14, 40)), (False, 13, (0, 1, 1), (70, 13, 12)),
                                                            # 23 = variant 3 with fused_overlap = 2 (boundary slabs, BCs and the whole exchange on the halo stream beside k_fused3d)
                                                            (True, 23, (1, 1, 1), (130, 14, 40)), (False, 23, (0, 1, 1), (70, 13, 12)), (True, 23, (1, 0, 1), (70, 13, 12))])
def test_solve_on_the_multi_gpu_path_matches_oracle_with_periodic_halo(jr, oracle, through_rccl, variant, periods, n):
    """The N > 1 code path of jrx_stokes3d_solve on one GPU: the grid is IGG-periodic in some dimensions, so the rank is its own
    neighbour there.  variant 2: split sweeps (boundary slabs first on the halo stream, interior concurrently on the compute stream,
    BCs + update_halo! behind the slabs; Stokes3D.jl:104-142).  variant 3: fused velocity+stress kernel, BCs, update_halo!, then the stress nodes next to a received plane redone (3: exchange behind the kernel;
    13: shell of tiles first on the halo stream, interior tiles concurrently; 23: boundary slabs + BCs + exchange on the halo stream beside the kernel).  Norms of the global count.  Expected = the CPU oracle's iteration followed by
    the same plane copies in numpy.  Tolerance 1e-12 of each field's max (observed: bit-identical)."""
    import ctypes as C
    import torch
    from justrelax_jl_amd import _lib, checks, halo
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    import justrelax_jl_amd.grid as g
    orc = oracle
    L = _lib.load()
    iters = 8
    s = jr.miniapps.random_fields3d(n, seed=11, iterMax=iters - 1, nout=4)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    g.init_global_grid(*n, periodx=periods[0], periody=periods[1], periodz=periods[2], rank=0, nprocs=1)
    ng = (g.nx_g(), g.ny_g(), g.nz_g())
    assert ng == tuple(n[d] - 2 * periods[d] for d in range(3))
    h = _lib.default_handle()
    try:
        halo.init_comm(h, self_rccl=through_rccl)
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant % 10))
        h.call("jrx_set_option", C.c_char_p(b"fused_overlap"), C.c_int64(variant // 10))
        stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
        r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
        got = download_stokes(stokes)
    finally:
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(0))
        h.call("jrx_set_option", C.c_char_p(b"fused_overlap"), C.c_int64(2))          # the default
        g.finalize_global_grid()
        g.init_global_grid(*n, rank=0, nprocs=1)
        halo.init_comm(h)          # back to a plain single-rank handle for the other tests
        g.finalize_global_grid()
    assert r.iter == iters

    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    b = s.flow_bcs
    pl = orc.params3d(n, s.grid._di["center"], s.dt, dict(r=s.pt.r, theta_dtau=s.pt.θ_dτ, eta_dtau=s.pt.ηdτ, eps_rel=1e-30, eps_abs=1e-30),
                      iterMax=iters - 1, nout=4, free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, ni_g=ng)

    def self_halo(arrs):
        new = _expected(arrs, n, periods, L)
        for a, m in zip(arrs, new):
            a[...] = m

    et = orc.compute_maxloc(ref["eta"])
    self_halo([et])
    errs = []
    for it in range(1, iters + 1):
        orc.stokes3d_iteration(ref, et, pl)
        self_halo([ref["Vx"], ref["Vy"], ref["Vz"]])
        if it % 4 == 0:
            ss = orc.residual_sumsq3d(ref, pl)
            cnt = [(ng[0] - 2) * (ng[1] - 1) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 2) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 1) * (ng[2] - 2),
                   ng[0] * ng[1] * ng[2]]
            errs.append(max(np.sqrt(ss[q]) / cnt[q] for q in range(4)))
    assert np.allclose(np.asarray(r.err_evo1), errs, rtol=1e-12), (r.err_evo1, errs)
    for k in ("P", "Vx", "Vy", "Vz", "txx", "tyy", "tzz", "txy", "txz", "tyz"):
        m = checks.interior_mask3d(k, ref[k].shape)
        scale = np.abs(ref[k]).max()
        d = np.abs(got[k] - ref[k])[m].max()
        assert d <= 1e-12 * scale, (k, d, scale)
    # ghost planes of V hold the wrapped interior planes
    for k in ("Vx", "Vz") if periods[0] else ():
        assert np.abs(got[k][0] - ref[k][0]).max() <= 1e-12 * np.abs(ref[k]).max()


def test_thermal3d_iterations_with_periodic_halo_match_oracle(jr, oracle):
    """update_halo!(thermal.T) inside the 3D heat-diffusion loop (DiffusionPT_solver.jl:110) on an IGG-periodic grid held by one rank:
    device iterations == oracle iteration + the same plane copies in numpy"""
    import torch
    from justrelax_jl_amd import _lib, halo, thermal as th
    from justrelax_jl_amd.arrays import from_numpy
    import justrelax_jl_amd.grid as g
    L = _lib.load()
    s = jr.miniapps.diffusion3d((14, 10, 9), iterMax=40, nout=20)
    n, periods = s.ni, (1, 1, 0)
    b = s.flow_bcs
    p = oracle.thermal_params3d(n, s.grid._di["center"], s.dt, 1e-30, iterMax=40, nout=20, no_flux=b.no_flux, constant_value=b.constant_value,
                                constant_flux=b.constant_flux, periodic=b.periodic)
    dev = torch.device("cuda", torch.cuda.current_device())
    thermal = jr.ThermalArrays(jr.AMDGPUBackend, n)
    for name in ("T", "H"):
        getattr(thermal, name).copy_(from_numpy(s.arrays[name], dev))
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-30)
    g.init_global_grid(*n, periodx=periods[0], periody=periods[1], periodz=periods[2], rank=0, nprocs=1)
    h = _lib.default_handle()
    try:
        halo.init_comm(h)
        r = jr.heatdiffusion_PT_(thermal, pt, b, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=40, nout=20, verbose=False))
    finally:
        g.finalize_global_grid()
        g.init_global_grid(*n, rank=0, nprocs=1)
        halo.init_comm(h)
        g.finalize_global_grid()
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    ref["Told"][...] = ref["T"]
    for it in range(40):
        oracle.thermal3d_iteration(ref, p)
        ref["T"][...] = _expected([ref["T"]], n, periods, L)[0]
    assert list(r.iter_count) == [20, 40]
    T = jr.to_numpy(thermal.T)
    inner = (slice(1, -1),) * 3
    assert np.abs(T[inner] - ref["T"][inner]).max() <= 1e-9 * np.abs(ref["T"]).max()
    assert np.abs(T[0, 1:-1, 1:-1] - ref["T"][0, 1:-1, 1:-1]).max() <= 1e-9 * np.abs(ref["T"]).max()       # x ghost plane: received, not a BC value
    assert np.abs(ref["T"][0, 1:-1, 1:-1] - ref["T"][1, 1:-1, 1:-1]).max() > 0.0                          # (no_flux would have copied plane 1)


@pytest.mark.parametrize("dim", [2, 3])
def test_vep_solve_with_periodic_halo_matches_oracle(jr, oracle, dim):
    """The multi-rank code path of the VEP drivers -- update_halo!(ητ), update_halo!(τ shear) and update_halo!(V) every iteration,
    norms of the global counts (Stokes2D.jl:655,757,784; Stokes3D.jl:515,578-580,596) -- on an IGG-periodic grid held by one rank,
    against the oracle driver with the same plane copies (orc_set_self_halo)."""
    import importlib
    import justrelax_jl_amd.grid as g
    from justrelax_jl_amd import _lib, halo
    L = oracle.lib()
    if dim == 3:
        tv = importlib.import_module("test_gpu_vep3d")
        s = jr.miniapps.shearband3d((20, 10, 9), iterMax=29, nout=10)
        periods, comps = (1, 0, 1), ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c")
    else:
        tv = importlib.import_module("test_gpu_vep2d")
        s = jr.miniapps.shearband2d(20, iterMax=29, nout=10)
        s.kwargs["iterMin"] = 5
        periods, comps = (1, 0, 0), ("xx", "yy", "xy", "xy_c")
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(13)
    for c in comps:                                  # pre-stress near yield: plasticity active
        s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
        s.arrays["t" + c][...] = s.arrays["to" + c]
    n3 = tuple(s.ni) + (1,) * (3 - dim)
    g.init_global_grid(*n3, periodx=periods[0], periody=periods[1], periodz=periods[2], rank=0, nprocs=1)
    ng = tuple(g.global_grid().n_g(d) for d in range(dim))
    h = _lib.default_handle()
    try:
        halo.init_comm(h)
        stokes, pr, ρg = tv._upload(jr, s)
        r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
        out = tv._download(jr, stokes)
    finally:
        g.finalize_global_grid()
        g.init_global_grid(*n3, rank=0, nprocs=1)
        halo.init_comm(h)
        g.finalize_global_grid()
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    rh = oracle.rheology_struct(s.extra["phases"])
    L.orc_set_self_halo(*periods)
    try:
        if dim == 3:
            r_ref = oracle.stokes3d_vep_solve(ref, rh, tv._params(oracle, s, ni_g=ng))
        else:
            r_ref = oracle.stokes2d_vep_solve(ref, rh, tv._vep_params(oracle, s, iterMin=5, ni_g=ng))
    finally:
        L.orc_set_self_halo(0, 0, 0)
    assert r.iter == r_ref["iter"] and len(r.err_evo1) == len(r_ref["err_evo1"]) >= 2
    assert np.allclose(r.err_evo1, r_ref["err_evo1"], rtol=1e-9)
    from justrelax_jl_amd.checks import interior_mask3d
    names = ("P", "Vx", "Vy", "txx", "tyy", "tII", "eta_vep") + (("Vz", "tzz", "tyz", "txz", "txy") if dim == 3 else ("txy",))
    for k in names:
        a, b = out[k], ref[k]
        m = interior_mask3d(k, b.shape) if dim == 3 else np.ones(b.shape, dtype=bool)
        assert np.abs(a - b)[m].max() <= 1e-9 * max(np.abs(b).max(), 1e-300), k
    assert (ref["eplxx"] != 0).any()
