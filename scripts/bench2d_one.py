#!/usr/bin/env python3
"""one 2D config for profiling: bench2d_one.py shearband|solcx n iters [KEY=INT ...]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench
h = _lib.default_handle(0)
kind, n, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    h.set_option(k, int(v))
r = (bench.cfg_shearband if kind == "shearband" else bench.cfg_solcx)(jr, h, n, iters)
print(json.dumps(r))
