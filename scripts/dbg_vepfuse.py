import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
from __graft_entry__ import load_package
jr = load_package()
import test_gpu_vep3d as T
from justrelax_jl_amd import _lib
h = _lib.default_handle()
ni, iters, nout = (20, 12, 10), int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 10
outs = []
for fuse in (0, 1, 1):
    h.set_option("vep3_fuse_pc", fuse)
    s = jr.miniapps.shearband3d(ni, iterMax=iters - 1, nout=nout)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
    rng = np.random.default_rng(3)
    for c in ("xx", "yy", "zz", "yz", "xz", "xy", "yz_c", "xz_c", "xy_c"):
        s.arrays["to" + c][...] = rng.uniform(-1.5, 1.5, size=s.arrays["to" + c].shape)
        s.arrays["t" + c][...] = s.arrays["to" + c]
    stokes, pr, ρg = T._upload(jr, s)
    r = jr.solve_(stokes, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None, kwargs=s.kwargs)
    outs.append(T._download(jr, stokes))
    print("fuse", fuse, "iter", r.iter, "graph replays", h.get_option("stat_graph_replays"))
for a, b, name in ((0, 1, "unfused vs fused"), (1, 2, "fused vs fused")):
    print(name)
    for k in outs[0]:
        d = outs[a][k] != outs[b][k]
        nanboth = np.isnan(outs[a][k]) & np.isnan(outs[b][k])
        d &= ~nanboth
        if d.any():
            idx = np.argwhere(d)
            print(f"  {k}: {d.sum()} of {d.size} differ, max abs {np.nanmax(np.abs(outs[a][k] - outs[b][k])):.3e}, first {idx[:4].tolist()}, nan {np.isnan(outs[a][k]).sum()} {np.isnan(outs[b][k]).sum()}")
