#!/usr/bin/env python3
"""Does the time of the 512^3 kernel depend on WHERE in the device's memory the arrays lie?  One process; per configuration: `lead` GiB of unused memory allocated first, then the
arrays (placement : chunk MiB), `ballast` MiB of unused memory behind every large array; everything is freed before the next configuration.
   probe_region.py [n=512] [configs lead:placement:chunk:ballast,...]"""
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
cfgs = sys.argv[2] if len(sys.argv) > 2 else ",".join(f"{l}:0:64:0" for l in (0, 16, 32, 48, 64, 96, 128, 160, 192, 0)) + "," + ",".join(f"0:0:64:{b}" for b in (512, 1024, 2048, 3072, 0))
torch.zeros(1, device="cuda")
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
for cfg in cfgs.split(","):
    lead, placement, chunk, ballast = (int(x) for x in cfg.split(":"))
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    leads = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(lead)]
    h = _lib.Handle(0)
    for k, v in (("operand_cache", 1), ("field_placement", placement), ("field_chunk_mib", chunk), ("field_ballast_mib", ballast)):
        h.set_option(k, v)
    arrays.use_library_arrays(h)
    t0 = time.time()
    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
    jr.flow_bcs_(st, bcs, handle=h)
    ητ = jr.fzeros((n, n, n), st.P.device)
    jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
    run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
    run(3)
    a = run(16)[4]
    b = run(16)[4]
    used = (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 30
    print(f"lead {lead:3d} GiB placement {placement} chunk {chunk:4d} MiB ballast {ballast:4d} MiB per array: {a:.3f} {b:.3f} ms   ({used:.0f} GiB in use, {time.time() - t0:.1f} s)", flush=True)
    del st, ρg, K, G, ητ, run, leads
    gc.collect()
    arrays.use_library_arrays(None)
    h.close()
    torch.cuda.empty_cache()
