#!/usr/bin/env python3
"""Hypothesis: a process runs the 512^3 kernel slowly when the driver backs its arrays with small physical fragments (the kernel keeps ~400 MiB of 22 arrays in flight: few TLB entries with
2 MiB fragments, thousands with 64 KiB ones).  This process allocates EVERY array of the run -- the caller's through hipExtMallocWithFlags(hipDeviceMallocContiguous) wrapped as torch
tensors, the library's scratch set through the tuning switch scratch_contiguous -- when argv[1] = 1, and as usual when 0, and prints the kernel time."""
import ctypes as C
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

contig = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
stagger = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # bytes (multiple of 256): array k starts (k mod 32) * stagger bytes into its allocation
torch.zeros(1, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
orig = arrays.fzeros
keep, failed = [], [0]


class Raw:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def fzeros_contig(shape, device, fill: float = 0.0):
    shape = tuple(int(s) for s in shape)
    cnt = 1
    for s_ in shape:
        cnt *= s_
    p = C.c_void_p()
    off = (len(keep) % 32) * stagger
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(cnt * 8 + off), C.c_uint(4))
    if rc != 0 or not p.value:
        failed[0] += 1
        return orig(shape, device, fill)
    raw = Raw(p.value + off, cnt)
    keep.append(raw)
    t = torch.as_tensor(raw, device=device)
    t.fill_(float(fill))
    return t.view(shape[::-1]).permute(*range(len(shape) - 1, -1, -1))


if contig:
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "__name__", "").startswith("justrelax_jl_amd") and getattr(mod, "fzeros", None) is orig:
            mod.fzeros = fzeros_contig
    if getattr(jr, "fzeros", None) is orig:
        jr.fzeros = fzeros_contig
h = _lib.default_handle(0)
h.set_option("scratch_contiguous", contig)
h.set_option("scratch_stagger", stagger)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
run(5)
k1 = run(41)[4]
h.set_option("zero_forces", 0)
k2 = run(41)[4]
print(f"contiguous {contig} stagger {stagger:9d} B: k_fused3d {k1:.3f} ms (with forces {k2:.3f} ms)   arrays through the contiguous path {len(keep)}, refused {failed[0]}", flush=True)
