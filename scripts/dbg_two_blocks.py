"""debug: per-component norms of a two-block run vs the expectation from the undecomposed run"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
jr = load_package()
import test_gpu_two_blocks as T
import _blocks as B
from justrelax_jl_amd import _lib
from justrelax_jl_amd.miniapps.common import download_stokes, upload_stokes
dims = tuple(int(c) for c in sys.argv[1]) if len(sys.argv) > 1 else (1, 1, 2)
n = (70, 13, 12)
kw = dict(iterMax=23, nout=8, verbose=False)
with T.TwoBlocks(n, dims) as tb:
    S = T._global_setup(jr, tb.ng, True, 23, 8)
    h0 = _lib.default_handle(); T._set(h0, kernel_variant=1)
    stokes, rg_, K, G = upload_stokes(S, jr.AMDGPUBackend)
    rg = jr.solve_(stokes, S.pt, S.grid, S.flow_bcs, rg_, K, G, S.dt, None, kwargs=kw)
    glob = download_stokes(stokes)
    res, outs = T._solve_blocks(jr, tb, S, "split_sweeps", kw)
ng = tb.ng
ss = np.zeros(4)
for r in range(2):
    co = B.coords_of(tb.carts[r])
    loc = {k: B.local_block(glob[k], n, ng, co) for k in ("Rx", "Ry", "Rz", "RP")}
    ss += [np.sum(loc["Rx"][1:-1, 1:-1, 1:-1] ** 2), np.sum(loc["Ry"][1:-1, 1:-1, 1:-1] ** 2), np.sum(loc["Rz"][1:-1, 1:-1, 1:-1] ** 2), np.sum(loc["RP"] ** 2)]
cnt = [(ng[0] - 2) * (ng[1] - 1) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 2) * (ng[2] - 1), (ng[0] - 1) * (ng[1] - 1) * (ng[2] - 2), ng[0] * ng[1] * ng[2]]
print("ng", ng, "want", [np.sqrt(ss[q]) / cnt[q] for q in range(4)])
for r in res:
    print("got ", [r.norm_Rx[-1], r.norm_Ry[-1], r.norm_Rz[-1], getattr(r, "norm_∇V")[-1]])
print("glob", [rg.norm_Rx[-1], rg.norm_Ry[-1], rg.norm_Rz[-1], getattr(rg, "norm_∇V")[-1]])
