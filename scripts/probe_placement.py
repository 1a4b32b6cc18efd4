#!/usr/bin/env python3
"""One process = one sample of the placement lottery.  Builds SolVi3D at n^3 with every array -- the caller's through
jrx_field_alloc (arrays.use_library_arrays), the library's second state set through the same pool -- under the placement given on the command line, and
prints the launch time of the headline kernel.

    probe_placement.py <mode> [n] [chunk_mib] [batch_mib] [va_align_mib] [shuffle] [iters] [arena_gib] [va_gap_mib]
    mode: torch (torch's allocator, the library's scratch by hipMalloc) | 0 (hipMalloc through the pool) | 1 (shuffled chunks) | 2 (contiguous)
"""
import ctypes as C
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

mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
chunk, batch, align, shuffle = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 64), (4, 0), (5, 0), (6, 1)))
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 41
arena, gap = (int(sys.argv[i]) if len(sys.argv) > i else 0 for i in (8, 9))
torch.zeros(1, device="cuda")
h = _lib.default_handle(0)
if mode != "torch":
    h.set_option("field_placement", int(mode))
    h.set_option("field_chunk_mib", chunk)
    h.set_option("field_batch_mib", batch)
    h.set_option("field_va_align_mib", align)
    h.set_option("field_shuffle", shuffle)
    h.set_option("field_arena_gib", arena)
    h.set_option("field_va_gap_mib", gap)
    arrays.use_library_arrays(h)
t0 = time.time()
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
torch.cuda.synchronize()
t_setup = time.time() - t0
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
run(5)
k1 = run(iters)[4]
k1b = run(iters)[4]
h.set_option("zero_forces", 0)
k2 = run(iters)[4]
stats = (C.c_int64 * 6)()
h.call("jrx_field_stats", stats)
print(f"mode {mode:>5} n {n} chunk {chunk:4d} MiB batch {batch:5d} align {align:3d} shuffle {shuffle} arena {arena} gap {gap:4d}: k_fused3d {k1:.3f} / {k1b:.3f} ms (with forces {k2:.3f} ms)   "
      f"setup {t_setup:.1f} s   pool: {stats[0]} arrays {stats[1] / 2**30:.1f} GiB, {stats[2]} chunks created in {stats[4] / 1e3:.0f} ms, mapped in {stats[5] / 1e3:.0f} ms", flush=True)
