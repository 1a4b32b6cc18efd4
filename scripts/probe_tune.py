#!/usr/bin/env python3
"""jrx_stokes3d_tune_placement on SolVi3D: what a search of `draws` draws finds, per chunk size (0 = every array one chunk), several processes' worth in one (new handle and arrays each).
   probe_tune.py [n=512] [draws=8] [chunks=0,64,1024] [repeats=2] [spread_draws=0] [pool_pct=70]"""
import gc
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
jr = load_package()
import torch  # noqa: E402
from justrelax_jl_amd import _lib, stokes, arrays  # noqa: E402
import justrelax_jl_amd.grid as grid  # noqa: E402
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 8
chunks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,64,1024").split(",")]
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 2
spread = int(sys.argv[5]) if len(sys.argv) > 5 else 0
pool = int(sys.argv[6]) if len(sys.argv) > 6 else 70
torch.zeros(1, device="cuda")
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
for rep in range(repeats):
    for chunk in chunks:
        h = _lib.Handle(0)
        for k, v in (("operand_cache", 1), ("field_placement", 1), ("field_chunk_mib", chunk), ("field_spread_draws", spread), ("field_pool_pct", pool)):
            h.set_option(k, v)
        arrays.use_library_arrays(h)
        st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
        jr.flow_bcs_(st, bcs, handle=h)
        ητ = jr.fzeros((n, n, n), st.P.device)
        jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
        t0 = time.time()
        ms, kept = stokes.tune_placement_(st, pt, geo, bcs, ρg, K, G, ητ, dt, draws, 12, handle=h)
        dt_s = time.time() - t0
        k = stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, 16, handle=h)
        print(f"n {n} chunk {chunk:4d} MiB spread {spread} pool {pool} %: as allocated {ms[0]:.3f}, draws " + " ".join(f"{x:.3f}" for x in ms[1:-1]) + f" -> {ms[-1]:.3f} ms per iteration ({kept} kept, {dt_s:.1f} s); kernel {k[4]:.3f} ms",
              flush=True)
        del st, ρg, K, G, ητ
        gc.collect()
        arrays.use_library_arrays(None)
        h.close()
        torch.cuda.empty_cache()
