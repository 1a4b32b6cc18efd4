"""hunt for the intermittent mismatch of tests/test_gpu_field_alloc.py::test_placement_search_...: the test's body over and over in ONE process, with other allocator traffic in between;
every mismatch is printed with where it is.   python scripts/stress_tune.py [rounds=30] [poison=0|1: memory the driver hands out next holds NaN / 1e300 / -7.25]"""
import ctypes as C, sys
import numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
import torch
from justrelax_jl_amd import _lib, arrays, checks, stokes
from justrelax_jl_amd.miniapps.common import download_stokes, stokes_field_names, upload_stokes, _get
from justrelax_jl_amd.arrays import from_numpy
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def leg(tune, chunk_mib, pool_pct, seed):
    h = _lib.Handle(0)
    try:
        h.set_option("field_placement", 1)
        h.set_option("field_chunk_mib", chunk_mib if tune else 2)
        h.set_option("field_pool_pct", pool_pct)
        arrays.use_library_arrays(h)
        s = jr.miniapps.random_fields3d((130, 96, 100), seed=seed, iterMax=40, nout=20)
        s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
        st, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
        if tune:
            ητ = jr.fzeros(s.ni, st.P.device)
            jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
            stokes.tune_placement_(st, s.pt, s.grid, s.flow_bcs, ρg, K, G, ητ, s.dt, 3, 4, handle=h)
            for name, path in stokes_field_names(3).items():
                if name in s.arrays:
                    _get(st, path).copy_(from_numpy(s.arrays[name], st.P.device))
                else:
                    _get(st, path).zero_()
            after = download_stokes(st)
            for name in s.arrays:
                if name in after and not np.array_equal(after[name], s.arrays[name]):
                    print("   the restored state differs from the initial one in", name, flush=True)
            del ητ
        r = jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=dict(s.kwargs, verbose=False), handle=h)
        out = (r, download_stokes(st))
        del st, ρg, K, G
        return out
    finally:
        arrays.use_library_arrays(None)
        h.close()


def traffic():
    """what the tests before it do: arrays of several sizes on chunks of several sizes, freed again"""
    for placement, chunk in ((1, 2), (1, 8), (2, 64), (0, 64)):
        h = _lib.Handle(0)
        h.set_option("field_placement", placement); h.set_option("field_chunk_mib", chunk)
        arrays.use_library_arrays(h)
        ts = [jr.fzeros(s, "cuda", fill=1.0) for s in ((257, 130, 67), (1200, 1100), (64, 64, 64), (300, 300, 30))]
        for t in ts:
            t.mul_(2.0)
        torch.cuda.synchronize()
        del ts, t
        arrays.use_library_arrays(None)
        h.close()
    x = [torch.empty(int(np.random.default_rng(7).integers(1, 40)) << 20, dtype=torch.uint8, device="cuda") for _ in range(6)]
    del x


def poison(kind):
    """memory the driver hands out next is not cleared: fill a good part of what is free with a pattern and give it back"""
    if kind == 0:
        return
    free = torch.cuda.mem_get_info()[0]
    n = int(min(free * 0.5, 60 * 2 ** 30) // 8 // 8)
    ts = [torch.full((n,), float("nan") if kind == 1 else 1.0e300 if kind == 2 else -7.25, dtype=torch.float64, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    del ts
    torch.cuda.empty_cache()


POISON = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for it in range(rounds):
    traffic()
    poison(POISON and 1 + it % 3)
    (ra, a) = leg(False, 0, 0, 5)
    for chunk_mib, pool_pct in ((0, 0), (10, 2)):
        poison(POISON and 1 + (it + 1) % 3)
        (rb, b) = leg(True, chunk_mib, pool_pct, 5)
        msg = []
        if ra.iter != rb.iter or list(ra.err_evo1) != list(rb.err_evo1):
            msg.append(f"iter {ra.iter} / {rb.iter}, err_evo1 {list(ra.err_evo1)} / {list(rb.err_evo1)}")
        for k in a:
            m = checks.interior_mask3d(k, a[k].shape)
            if not np.array_equal(a[k][m], b[k][m], equal_nan=True):
                d = np.abs(a[k] - b[k]); msg.append(f"{k}: {int((a[k] != b[k]).sum())} entries differ, max {np.nanmax(d):.3e}, first at {np.argwhere(a[k] != b[k])[0].tolist()}")
        if msg:
            bad += 1
            print(f"round {it} (chunk {chunk_mib}, pool {pool_pct}): MISMATCH: " + "; ".join(msg[:6]), flush=True)
print(f"{rounds} rounds, {bad} mismatches", flush=True)
