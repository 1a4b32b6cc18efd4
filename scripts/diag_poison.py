"""Round 6 diagnosis: which path of a coupled-blocks solve reads library-owned memory that nothing has written?  Every library allocation is filled with NaNs
("scratch_poison"); the same solve on two blocks with / without the poison and with / without the placement pool, for several pipelines and splits."""
import sys, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
jr = load_package()
import test_gpu_two_blocks as T
import _blocks as B
import justrelax_jl_amd.grid as g
from justrelax_jl_amd import arrays, halo
from justrelax_jl_amd.miniapps.common import Setup, download_stokes, upload_stokes


def run(dims, n, pipeline, pooled, poison, dt=0.25, iters=30):
    with T.TwoBlocks(n, dims) as tb:
        S = T._global_setup(jr, tb.ng, False, iters, 10, dt=dt)
        g.init_global_grid(*n, dimx=dims[0], dimy=dims[1], dimz=dims[2], rank=0, nprocs=len(tb.handles))
        try:
            grid = jr.Geometry(n, S.extra["li"])
            ups = []
            for r, h in enumerate(tb.handles):
                T._set(h, **T.PIPELINES[pipeline])
                h.set_option("scratch_poison", poison)
                if pooled:
                    h.set_option("field_placement", 1); h.set_option("field_chunk_mib", 128); h.set_option("field_pool_pct", 1)
                    arrays.use_library_arrays(h)
                loc = Setup(ni=n, arrays={k: B.local_block(v, n, tb.ng, B.coords_of(tb.carts[r])) for k, v in S.arrays.items()})
                ups.append(upload_stokes(loc, jr.AMDGPUBackend))
                arrays.use_library_arrays(None)
            kw = dict(iterMax=iters, nout=10, verbose=False)
            solve = lambda r: jr.solve_(ups[r][0], S.pt, grid, S.flow_bcs, ups[r][1], ups[r][2], ups[r][3], S.dt, None, kwargs=kw, handle=tb.handles[r])
            res = halo.run_ranks([(lambda r=r: solve(r)) for r in range(len(tb.handles))])
            out = [download_stokes(u[0]) for u in ups]
            del ups
            return res, out
        finally:
            arrays.use_library_arrays(None)
            g.finalize_global_grid()


n = (130, 96, 100)
for dims in ((2, 1, 1), (1, 1, 2)):
    for dt in (0.25, float("inf")):
        for pipeline in ("fused", "fused_early", "fused_inkernel", "split_sweeps"):
            base = None
            for pooled, poison in ((0, 0), (0, 1), (1, 0), (1, 1)):
                tag = f"dims {dims} dt {dt} {pipeline:15s} pooled {pooled} poison {poison}"
                try:
                    res, out = run(dims, n, pipeline, pooled, poison, dt)
                    if base is None:
                        base = out
                        print(tag, "-> ok (reference)", flush=True)
                    else:
                        bad = [(r, k) for r in range(len(out)) for k in T.STATE if not np.array_equal(out[r][k], base[r][k], equal_nan=True)]
                        print(tag, "-> ok, same bits" if not bad else f"-> DIFFERENT: {bad[:6]}", flush=True)
                except Exception as e:
                    print(tag, f"-> {type(e).__name__}: {e}", flush=True)
