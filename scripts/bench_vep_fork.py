#!/usr/bin/env python3
"""A/B of the 3D VEP driver's forked centre pass (tuning switch vep3_fork): shear band 256^3 (and 160^3), alternating, same process and allocations.
    python3 scripts/bench_vep_fork.py [n=256] [rounds=3]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench_extras as bench
from __graft_entry__ import load_package

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
jr = load_package()
from justrelax_jl_amd import _lib
h = _lib.default_handle(0)
for r in range(rounds):
    for fork in (1, 0):
        h.set_option("vep3_fork", fork)
        out = bench.cfg_shearband3d(jr, h, n=n, iters=60)
        print(f"n={n} vep3_fork={fork}: {out['it_per_s']:.1f} it/s  frac {out['frac_at_needed_bytes']:.3f}", flush=True)
h.set_option("vep3_fork", 1)
