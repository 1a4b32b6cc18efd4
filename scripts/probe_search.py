#!/usr/bin/env python3
"""Placement SEARCH in one process (jrx_tuning_field_reroll / _undo / _keep): (A) complete draws, keep the best; (B) one array at a time, keep a draw only if the kernel got faster.
   probe_search.py [n=512] [chunk_mib=64] [draws=8] [sweeps=2] [skew=0]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
draws = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
skew = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.zeros(1, device="cuda")
h = _lib.Handle(0)
for k, v in (("operand_cache", 1), ("field_placement", 1), ("field_chunk_mib", chunk), ("field_skew_bytes", skew)):
    h.set_option(k, v)
arrays.use_library_arrays(h)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)


def probe(k=12):
    run(2)
    return run(k)[4]


call = lambda f, p=None: (torch.cuda.synchronize(), h.call(f, C.c_void_p(p or 0)))
run(3)
best = probe(16)
print(f"n {n} chunk {chunk} MiB skew {skew}: as allocated {best:.3f} ms", flush=True)
t0 = time.time()
traj = []
for d in range(draws):
    call("jrx_tuning_field_reroll")
    t = probe()
    if t < best:
        best = t
        call("jrx_tuning_field_keep")
        traj.append(f"{t:.3f}*")
    else:
        call("jrx_tuning_field_undo")
        traj.append(f"{t:.3f}")
print(f"(A) {draws} complete draws, * = kept ({time.time() - t0:.1f} s): " + " ".join(traj) + f" -> {probe(16):.3f} ms", flush=True)
cnt = C.c_int64()
ptrs, nbytes = (C.c_void_p * 256)(), (C.c_int64 * 256)()
h.call("jrx_field_list", C.c_int64(256), ptrs, nbytes, C.byref(cnt))
big = [ptrs[i] for i in range(cnt.value) if nbytes[i] >= n ** 3 * 8]
for s in range(sweeps):
    t0 = time.time()
    kept = 0
    for p in big:
        call("jrx_tuning_field_reroll", p)
        t = probe()
        if t < best - 0.004:
            best = t
            kept += 1
            call("jrx_tuning_field_keep", p)
        else:
            call("jrx_tuning_field_undo", p)
    print(f"(B) sweep {s}: {len(big)} arrays one at a time, {kept} draws kept ({time.time() - t0:.1f} s) -> {probe(16):.3f} ms", flush=True)
print(f"final {probe(16):.3f} ms; again {probe(16):.3f} ms", flush=True)
