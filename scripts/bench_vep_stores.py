#!/usr/bin/env python3
"""A/B of the skipped output-only stores of the VEP loops (tuning switch vep_store_all), same process: 3D shear band 256^3 and 2D shear band 1024^2"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
import bench_extras as bench
h = _lib.default_handle(0)
for rep in range(3):
    for allst in (0, 1):
        h.set_option("vep_store_all", allst)
        print(json.dumps({"vep_store_all": allst, "vep3d_256_it_per_s": round(bench.cfg_shearband3d(jr, h)["it_per_s"], 1),
                          "shearband2d_1024_it_per_s": round(bench.cfg_shearband(jr, h)["it_per_s"], 1)}), flush=True)
h.set_option("vep_store_all", 0)
