"""two coupled 512^3 blocks on one device (in-process transport, default pipeline), a few iterations, for rocprofv3 --kernel-trace: which kernels run per iteration and for how long
   python3 scripts/trace_two_blocks.py [x|z] [coupled=1 | 0 | 2 = coupled, then uncoupled in the same process] [iters=10]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
import torch
import justrelax_jl_amd.grid as grid
from justrelax_jl_amd import _lib, halo, stokes
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
split = sys.argv[1] if len(sys.argv) > 1 else "x"
coupled = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = 512
dims = {"x": (2, 1, 1), "y": (1, 2, 1), "z": (1, 1, 2)}[split]
dev = torch.cuda.current_device()
hs = [_lib.Handle(dev) for _ in range(2)]
blocks = []
for r in range(2):
    grid.finalize_global_grid()
    grid.init_global_grid(n, n, n, rank=r, nprocs=2, dimx=dims[0], dimy=dims[1], dimz=dims[2])
    hs[r].set_option("operand_cache", 1)
    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
    jr.flow_bcs_(st, bcs, handle=hs[r])
    ητ = jr.fzeros((n, n, n), st.P.device)
    jr.compute_maxloc_(ητ, st.viscosity.η, handle=hs[r])
    blocks.append((st, pt, geo, bcs, ρg, K, G, ητ, dt))
if coupled:
    halo.init_comm_local(hs, halo.make_carts((n, n, n), dims))
    halo.run_ranks([(lambda r=r: halo.update_halo_(blocks[r][0].V.Vx, blocks[r][0].V.Vy, blocks[r][0].V.Vz, blocks[r][7], ni=(n, n, n), handle=hs[r])) for r in range(2)])
fns = lambda m: [(lambda r=r: stokes.iterate_timed_(*blocks[r], m, handle=hs[r])) for r in range(2)]
halo.run_ranks(fns(6))
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
halo.run_ranks(fns(iters))
torch.cuda.synchronize()
print(f"split {split} coupled {coupled}: {2 * iters / (time.perf_counter() - t0):.1f} block-it/s", flush=True)
if coupled == 2:        # the same blocks again without the communicator, in the same process (same arrays, same placement)
    for h in hs:
        h.call("jrx_comm_destroy")
    halo.run_ranks(fns(6))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    halo.run_ranks(fns(iters))
    torch.cuda.synchronize()
    print(f"split {split} uncoupled, same process: {2 * iters / (time.perf_counter() - t0):.1f} block-it/s", flush=True)
