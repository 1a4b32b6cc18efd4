"""Round 6 diagnosis, part 3: jrx_stokes3d_solve on two coupled blocks with every library allocation poisoned fails with NaN(s) at the first check -- which arrays hold NaNs then?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
jr = load_package()
import test_gpu_two_blocks as T
import _blocks as B
import justrelax_jl_amd.grid as g
from justrelax_jl_amd import halo
from justrelax_jl_amd.miniapps.common import Setup, download_stokes, upload_stokes


def run(dims, n, pipeline, poison, dt, iters, nout):
    with T.TwoBlocks(n, dims) as tb:
        S = T._global_setup(jr, tb.ng, False, iters, nout, dt=dt)
        g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=dims[2], rank=0, nprocs=len(tb.handles))
        try:
            grid = jr.Geometry(n, S.extra["li"])
            ups = []
            for r, h in enumerate(tb.handles):
                T._set(h, **T.PIPELINES[pipeline])
                h.set_option("scratch_poison", poison)
                loc = Setup(ni=n, arrays={k_: B.local_block(v, n, tb.ng, B.coords_of(tb.carts[r])) for k_, v in S.arrays.items()})
                ups.append(upload_stokes(loc, jr.AMDGPUBackend))
            kw = dict(iterMax=iters, nout=nout, verbose=False)
            def solve(r):
                try:
                    return jr.solve_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][1], ups[r][2], ups[r][3], S.dt, None, kwargs=kw, handle=tb.handles[r])
                except Exception as e:
                    return e
            res = halo.run_ranks([(lambda r=r: solve(r)) for r in range(2)])
            return res, [download_stokes(u[0]) for u in ups]
        finally:
            g.finalize_global_grid()


n = (130, 96, 100)
for dims, pipeline, iters, nout, mask in (((2, 1, 1), "fused", 5, 5, 1), ((2, 1, 1), "fused", 5, 5, 2), ((2, 1, 1), "fused", 10, 5, 1), ((2, 1, 1), "fused", 10, 5, 2),
                                          ((2, 1, 1), "fused_early", 10, 5, 1), ((2, 1, 1), "fused_early", 10, 5, 2), ((1, 1, 2), "fused", 30, 10, 1), ((1, 1, 2), "fused", 30, 10, 2)):
    res, out = run(dims, n, pipeline, mask, 0.25, iters, nout)
    print(f"dims {dims} {pipeline} iterMax {iters} nout {nout} poison mask {mask}: {[str(r)[:60] for r in res]}", flush=True)
    for r in range(2):
        for name in ("P", "txx", "txy", "txz", "tyz", "Vx", "Vy", "Vz"):
            a = out[r][name]
            bad = np.argwhere(np.isnan(a))
            if len(bad):
                print(f"   rank {r} {name:6s} shape {a.shape}: {len(bad)} NaNs, index box {bad.min(axis=0).tolist()} .. {bad.max(axis=0).tolist()}", flush=True)
