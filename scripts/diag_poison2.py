"""Round 6 diagnosis, part 2: where are the NaNs?  Two coupled blocks, every library allocation poisoned, k iterations without norm checks (jrx_stokes3d_iterate_timed), then every
array of the state is searched for NaNs."""
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
from justrelax_jl_amd import halo, stokes
from justrelax_jl_amd.miniapps.common import Setup, download_stokes, upload_stokes


def run(dims, n, pipeline, poison, dt, k):
    with T.TwoBlocks(n, dims) as tb:
        S = T._global_setup(jr, tb.ng, False, 30, 10, dt=dt)
        g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=dims[2], rank=0, nprocs=len(tb.handles))
        try:
            grid = jr.Geometry(n, S.extra["li"])
            ups, ets = [], []
            for r, h in enumerate(tb.handles):
                T._set(h, **T.PIPELINES[pipeline])
                h.set_option("scratch_poison", poison)
                loc = Setup(ni=n, arrays={k_: B.local_block(v, n, tb.ng, B.coords_of(tb.carts[r])) for k_, v in S.arrays.items()})
                ups.append(upload_stokes(loc, jr.AMDGPUBackend))
            for r, h in enumerate(tb.handles):
                et = jr.fzeros(n, ups[r][0].P.device)
                jr.compute_maxloc_(et, ups[r][0].viscosity.η, handle=h)
                ets.append(et)
            halo.run_ranks([(lambda r=r: halo.update_halo_(ups[r][0].V.Vx, ups[r][0].V.Vy, ups[r][0].V.Vz, ets[r], ni=n, handle=tb.handles[r])) for r in range(2)])
            it = lambda r: stokes.iterate_timed_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][1], ups[r][2], ups[r][3], ets[r], S.dt, k, handle=tb.handles[r])
            halo.run_ranks([(lambda r=r: it(r)) for r in range(2)])
            return [download_stokes(u[0]) for u in ups]
        finally:
            g.finalize_global_grid()


n = (130, 96, 100)
for dims in ((2, 1, 1), (1, 1, 2)):
    for dt in (0.25,):
        for pipeline in ("fused", "fused_early"):
            for k in (1, 2, 3, 4):
                ref = run(dims, n, pipeline, 0, dt, k)
                out = run(dims, n, pipeline, 1, dt, k)
                for r in range(2):
                    for name in sorted(out[r]):
                        a, b = out[r][name], ref[r][name]
                        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
                        if len(bad):
                            lo, hi = bad.min(axis=0), bad.max(axis=0)
                            print(f"dims {dims} {pipeline:12s} k {k} rank {r} {name:6s} shape {a.shape}: {len(bad)} entries differ, index box {lo.tolist()} .. {hi.tolist()}, NaNs {int(np.isnan(a).sum())}", flush=True)
                print(f"dims {dims} {pipeline} k {k}: done", flush=True)
