"""Round 6: 3D heat diffusion at 256^3, array form and phase-ratio form: non-temporal stores of the new (T, qT) set (tuning switch thermal_nt) against plain stores."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench_extras as X
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib
h = _lib.default_handle()
for rep in range(3):
    for nt in (0, 1):
        h.set_option("thermal_nt", nt)
        a = X.cfg_thermal3d(jr, h, n=256, iters=400)
        b = X.cfg_thermal3d_phases(jr, h, n=256, iters=200)
        print(f"thermal_nt {nt}: array form {a['it_per_s']:.1f} it/s   phase-ratio form {b['it_per_s']:.1f} it/s", flush=True)
h.set_option("thermal_nt", 0)
