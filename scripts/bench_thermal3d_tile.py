"""Round 6: the 3D heat-diffusion iteration at 256^3 (array form), row-segment form against the 64 x TY tiles with y neighbours through LDS (tuning switch thermal_tile) and XCD band widths."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench_extras as X
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
h = _lib.default_handle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for rep in range(2):
    for tile, xg in ((0, 8), (4, 8), (4, 1), (4, 4), (8, 8), (8, 2), (8, 4)):
        h.set_option("thermal_tile", tile); h.set_option("thermal_xg", xg)
        try:
            r = X.cfg_thermal3d(jr, h, n=n, iters=400)
            print(f"n {n} thermal_tile {tile} thermal_xg {xg}: {r['it_per_s']:.1f} it/s  frac at needed bytes {r.get('frac_at_needed_bytes', 0):.3f}", flush=True)
        except Exception as e:
            print(f"n {n} thermal_tile {tile} thermal_xg {xg}: {type(e).__name__}: {e}", flush=True)
h.set_option("thermal_tile", 0); h.set_option("thermal_xg", 8)
