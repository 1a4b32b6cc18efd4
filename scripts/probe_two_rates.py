#!/usr/bin/env python3
"""One process: the kernel time of the headline form at 512^3, then plain torch streaming probes on the same device (a 1 GiB copy, a 3-stream add, an 8 GiB copy), so that a process that
runs k_fused3d at the slow rate can be asked whether everything that streams is slow in it (scripts/gpu_r04_two_rates.sh runs several processes and samples rocm-smi beside them)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
jr = load_package()
import torch  # noqa: E402
from justrelax_jl_amd import _lib, stokes  # noqa: E402
import justrelax_jl_amd.grid as grid  # noqa: E402
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device  # noqa: E402

n = 512
h = _lib.default_handle(0)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
run(5)
k1 = run(41)[4]


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a, b, c = st.P, st.τ.xx, st.τ.yy            # three of the run's own 1 GiB arrays (contents do not matter any more)
t_copy = timed(lambda: c.copy_(a))
t_add = timed(lambda: torch.add(a, b, out=c))
big = torch.empty(2 ** 30, dtype=torch.float64, device=a.device)       # 8 GiB
big2 = torch.empty(2 ** 30, dtype=torch.float64, device=a.device)
t_big = timed(lambda: big2.copy_(big), reps=5)
k2 = run(41)[4]
gb = 2 ** 30 / 1e9
print(f"k_fused3d {k1:.3f} / {k2:.3f} ms   copy 1 GiB {t_copy:.4f} ms = {2 * gb / t_copy:.2f} TB/s   add 3 x 1 GiB {t_add:.4f} ms = {3 * gb / t_add:.2f} TB/s   copy 8 GiB {t_big:.3f} ms = {16 * gb / t_big:.2f} TB/s", flush=True)
