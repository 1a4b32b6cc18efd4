"""Graph replay of the launch-bound 3D loops: on small grids -- the sizes the reference's own 3D tests run at
(test/test_stokes_solvi3D.jl:25-55: 16^3; test_diffusion3D.jl: 32^3; test_shearband3D_MPI.jl) -- runs of unobserved iterations of
jrx_stokes3d_solve (un-fused sweeps), jrx_stokes3d_vep_solve and jrx_heatdiffusion_PT3d replay as captured hipGraphs.  Same kernels in the
same order: every output must be bit-identical to plain launches, the iteration counts and norms equal, and the counters must show that the
graphs really ran."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _opt(h, key, v=None):
    if v is not None:
        h.call("jrx_set_option", C.c_char_p(key), C.c_int64(v))
    out = C.c_int64(0)
    h.call("jrx_get_option", C.c_char_p(key), C.byref(out))
    return out.value


@pytest.mark.parametrize("n", [(16, 16, 16), (33, 20, 18), (40, 48, 44)])
def test_stokes3d_graph_replay_changes_nothing(jr, oracle, n):
    from justrelax_jl_amd import _lib, checks
    from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
    h = _lib.default_handle()
    outs = []
    try:
        for g in (0, 1):
            _opt(h, b"loop_graphs", g)
            s = jr.miniapps.random_fields3d(n, seed=3, iterMax=149, nout=50)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            stokes, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
            r0 = _opt(h, b"stat_graph_replays")
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
            outs.append((r, download_stokes(stokes), _opt(h, b"stat_graph_replays") - r0))
    finally:
        _opt(h, b"loop_graphs", 1)
    (ra, a, na), (rb, b, nb) = outs
    assert na == 0 and nb >= 6, (na, nb)              # 150 iterations, checks at 50 / 100 / 150: 3 runs of 49 unobserved iterations = 3 x 3 graphs of 16
    assert ra.iter == rb.iter == 150 and list(ra.err_evo1) == list(rb.err_evo1)
    for k in a:         # the edge / corner ghosts of V and U are written racily by flow_bcs! (and never read): excluded, as everywhere
        m = checks.interior_mask3d(k, a[k].shape)
        assert np.array_equal(a[k][m], b[k][m], equal_nan=True), k
    # and against the oracle (the graphs replay the reference's iteration)
    ref = {k: v.copy(order="F") for k, v in s.arrays.items()}
    r_ref = oracle.stokes3d_solve(ref, checks.oracle_params3d(oracle, s))
    assert r_ref["iter"] == 150
    for k in ("P", "Vx", "Vy", "Vz", "txx", "txy", "tyz"):
        m = checks.interior_mask3d(k, ref[k].shape)
        assert np.abs(b[k] - ref[k])[m].max() <= 1e-9 * np.abs(ref[k]).max(), k


def test_vep3d_graph_replay_changes_nothing(jr):
    import test_gpu_vep3d as tv
    from justrelax_jl_amd import _lib
    h = _lib.default_handle()
    outs = []
    try:
        for g in (0, 1):
            _opt(h, b"loop_graphs", g)
            s = jr.miniapps.shearband3d((24, 20, 18), iterMax=119, nout=40)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
            rng = np.random.default_rng(3)
            for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):
                s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
                s.arrays["t" + c][...] = s.arrays["to" + c]
            stokes, pr, ρg = tv._upload(jr, s)
            r0 = _opt(h, b"stat_graph_replays")
            r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
            outs.append((r, tv._download(jr, stokes), _opt(h, b"stat_graph_replays") - r0))
    finally:
        _opt(h, b"loop_graphs", 1)
    (ra, a, na), (rb, b, nb) = outs
    assert na == 0 and nb >= 4, (na, nb)
    assert ra.iter == rb.iter == 120 and list(ra.err_evo1) == list(rb.err_evo1)
    assert (a["eplxx"] != 0).any()
    from justrelax_jl_amd.checks import interior_mask3d
    for k in a:
        m = interior_mask3d(k, a[k].shape)
        assert np.array_equal(a[k][m], b[k][m], equal_nan=True), k


def test_thermal3d_graph_replay_changes_nothing(jr):
    import torch
    from justrelax_jl_amd import _lib
    from justrelax_jl_amd.arrays import from_numpy
    h = _lib.default_handle()
    dev = torch.device("cuda", torch.cuda.current_device())
    outs = []
    try:
        for g in (0, 1):
            _opt(h, b"loop_graphs", g)
            s = jr.miniapps.diffusion3d((32, 24, 20), iterMax=150, nout=50)
            thermal = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
            for name in ("T", "H"):
                getattr(thermal, name).copy_(from_numpy(s.arrays[name], dev))
            K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
            pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-30)
            r0, f0 = _opt(h, b"stat_graph_replays"), _opt(h, b"stat_thermal_fused")
            r = jr.heatdiffusion_PT_(thermal, pt, s.flow_bcs, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=150, nout=50, verbose=False))
            outs.append((r, {k: jr.to_numpy(getattr(thermal, k)) for k in ("T", "qTx", "qTy", "qTz", "qTx2", "ResT", "ΔT")},
                         _opt(h, b"stat_graph_replays") - r0, _opt(h, b"stat_thermal_fused") - f0))
    finally:
        _opt(h, b"loop_graphs", 1)
    (ra, a, na, fa), (rb, b, nb, fb) = outs
    assert na == 0 and nb >= 3 and fa == fb == 147, (na, nb, fa, fb)        # 150 iterations, 3 of them observed
    assert list(ra.iter_count) == list(rb.iter_count) == [50, 100, 150] and list(ra.norm_ResT) == list(rb.norm_ResT)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
