#!/usr/bin/env python3
"""Is the rate of the 512^3 kernel a property of the PROCESS or of each set of allocations?  One process builds SolVi3D several times (the previous set is still held while the next one
is allocated, so the allocator cannot hand the same memory back) and times the headline kernel on each set.  argv[1]: torch | 0 | 1 (placement), argv[2]: n, argv[3]: sets, argv[4]: chunk MiB"""
import sys
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
sets = int(sys.argv[3]) if len(sys.argv) > 3 else 4
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 64
torch.zeros(1, device="cuda")
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
held, out = [], []
for t in range(sets):
    h = _lib.Handle(0)                   # a handle of its own per set: its second state set is part of the placement
    h.set_option("operand_cache", 1)
    if mode != "torch":
        h.set_option("field_placement", int(mode))
        h.set_option("field_chunk_mib", chunk)
        arrays.use_library_arrays(h)
    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
    jr.flow_bcs_(st, bcs, handle=h)
    ητ = jr.fzeros((n, n, n), st.P.device)
    jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
    run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
    run(5)
    a = run(21)[4]
    b = run(21)[4]
    out.append((a, b))
    held.append((h, st, ρg, K, G, ητ))
    if len(held) > 1:                    # two sets alive at most: the one before the last goes now, after the next one was placed
        old = held.pop(0)
        arrays.use_library_arrays(None)
        del old
        torch.cuda.empty_cache()
print(f"mode {mode} n {n} chunk {chunk}: " + "  ".join(f"{a:.3f}/{b:.3f}" for a, b in out), flush=True)
