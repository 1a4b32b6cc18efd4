#!/usr/bin/env python3
"""SolCx (2D Stokes, dt = Inf) it/s over grid sizes: the control-flow one-launch iteration / two-kernel loop as shipped before (fused2d_batch = 0), and the batched one-launch
iteration forced at every size (fused2d_batch = 1, fused2d_max_nodes = 10^9), with and without its viscous-limit instantiation"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench_extras as bench
h = _lib.default_handle(0)
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 384, 512, 768, 1024, 1280, 1536, 2048]:
    iters = max(400, min(8000, int(2e9 / n ** 2)))
    row = {"n": n, "iters": iters}
    for rep in range(2):
        for name, opts in (("before", dict(fused2d_batch=0)), ("two_kernels", dict(kernel_variant=2)), ("batched", dict(fused2d_batch=1, fused2d_max_nodes=10 ** 9)),
                           ("batched_general_form", dict(fused2d_batch=1, fused2d_max_nodes=10 ** 9, viscous_limit=0))):
            for k, v in dict(fused2d_batch=1, fused2d_max_nodes=200000, viscous_limit=1, kernel_variant=0).items():
                h.set_option(k, v)
            for k, v in opts.items():
                h.set_option(k, v)
            row.setdefault(name, []).append(round(bench.cfg_solcx(jr, h, n, iters)["it_per_s"], 0))
    print(json.dumps(row), flush=True)
