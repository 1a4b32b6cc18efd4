#!/usr/bin/env python3
"""it/s of the 2D configs of BASELINE.json on one GPU: SolCx 512^2 (config 2), shear band 1024^2 (config 5),
thermal diffusion 256^2 (config 1) -- the `other_configs` legs of bench.py run on their own.  One JSON line per config."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib

if __name__ == "__main__":
    h = _lib.default_handle(0)
    for fn in (bench.cfg_solcx, bench.cfg_shearband, bench.cfg_thermal2d):
        print(json.dumps(fn(jr, h)), flush=True)
