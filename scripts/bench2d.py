#!/usr/bin/env python3
"""it/s of the 2D configs of BASELINE.json on one GPU: SolCx 512^2 (config 2), shear band 1024^2 (config 5),
thermal diffusion 256^2 (config 1) -- the `other_configs` legs of bench.py run on their own.  One JSON line per config."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench_extras as bench
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib

if __name__ == "__main__":
    # usage: bench2d.py [solcx|shearband|thermal ...] [KEY=INT ...]   (library options; default: all three configs)
    import ctypes as C
    h = _lib.default_handle(0)
    table = dict(solcx=bench.cfg_solcx, shearband=bench.cfg_shearband, thermal=bench.cfg_thermal2d)
    names = [a for a in sys.argv[1:] if "=" not in a] or list(table)
    for kv in (a for a in sys.argv[1:] if "=" in a):
        k, v = kv.split("=")
        h.set_option(k, int(v))
        print(f"# option {k} = {v}", flush=True)
    for n in names:
        print(json.dumps(table[n](jr, h)), flush=True)
