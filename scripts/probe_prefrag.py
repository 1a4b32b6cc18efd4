#!/usr/bin/env python3
"""The slow rate is the physically contiguous placement (profiles/r04_alloc_stagger.txt).  Can a process make sure its arrays are NOT backed by long contiguous runs?  Before building the
problem this process fills `fill_gb` of device memory with blocks of `chunk_mb` MiB, returns every other one to the driver (torch.cuda.empty_cache), builds the problem in the holes and
frees the rest.   python3 scripts/probe_prefrag.py chunk_mb [fill_gb]     (chunk_mb = 0: build the problem as usual)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
jr = load_package()
import time  # noqa: E402
import torch  # noqa: E402
from justrelax_jl_amd import _lib, stokes  # noqa: E402
import justrelax_jl_amd.grid as grid  # noqa: E402
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device  # noqa: E402

chunk_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fill_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
n = 512
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
t0 = time.perf_counter()
keepers = []
if chunk_mb > 0:
    free, _ = torch.cuda.mem_get_info(dev)
    total = int(min(fill_gb * 2 ** 30, 0.8 * free))
    nchunk = total // (chunk_mb << 20)
    blocks = [torch.empty(chunk_mb << 20, dtype=torch.uint8, device=dev) for _ in range(nchunk)]
    keepers = blocks[0::2]
    del blocks
    torch.cuda.empty_cache()          # every other block goes back to the driver: free device memory is now a comb of chunk-sized holes
prep = time.perf_counter() - t0
h = _lib.default_handle(0)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
run(5)
del keepers
torch.cuda.empty_cache()
k1 = run(41)[4]
print(f"chunk {chunk_mb:4d} MiB: k_fused3d {k1:.3f} ms   (comb of holes prepared in {prep:.2f} s)", flush=True)
