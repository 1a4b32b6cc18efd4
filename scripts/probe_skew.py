#!/usr/bin/env python3
"""Does the time of the 512^3 kernel depend on where element i of the different arrays sits inside its 2 MiB page ("field_skew_bytes"), and how does it spread over physical
placements (complete re-rolls, which since the translation flush of csrc/fieldpool.hip do take effect)?  One process; per configuration: new handle, new arrays, probes.
   probe_skew.py [n=512] [rolls=6] [configs: placement:chunk:skew:mod,...]"""
import ctypes as C
import gc
import statistics
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
rolls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfgs = sys.argv[3] if len(sys.argv) > 3 else "0:64:0:32,0:64:4096:32,1:64:0:32,1:64:4096:32,1:64:256:32,1:64:69632:32,1:2:0:32,1:2:4096:32,1:1024:0:32,1:1024:4096:32,0:64:0:32"
torch.zeros(1, device="cuda")
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
for cfg in cfgs.split(","):
    placement, chunk, skew, mod = (int(x) for x in cfg.split(":"))
    h = _lib.Handle(0)
    h.set_option("operand_cache", 1)
    h.set_option("field_placement", placement)
    h.set_option("field_chunk_mib", chunk)
    h.set_option("field_skew_bytes", skew)
    h.set_option("field_skew_mod", mod)
    arrays.use_library_arrays(h)
    t0 = time.time()
    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
    jr.flow_bcs_(st, bcs, handle=h)
    ητ = jr.fzeros((n, n, n), st.P.device)
    jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
    run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
    run(3)
    out = [run(16)[4]]
    if placement == 1:
        for r in range(rolls):
            torch.cuda.synchronize()
            h.call("jrx_tuning_field_reroll", C.c_void_p(0))
            run(2)
            out.append(run(16)[4])
    print(f"placement {placement} chunk {chunk:4d} MiB skew {skew:7d} B mod {mod:2d}: min {min(out):.3f} median {statistics.median(out):.3f} max {max(out):.3f} ms  (" + " ".join(f"{x:.3f}" for x in out)
          + f")  [{time.time() - t0:.1f} s]", flush=True)
    del st, ρg, K, G, ητ, run
    gc.collect()
    arrays.use_library_arrays(None)
    h.close()
    torch.cuda.empty_cache()
