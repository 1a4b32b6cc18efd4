import sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from __graft_entry__ import load_package
jr = load_package()
from justrelax_jl_amd import _lib, checks
from justrelax_jl_amd.miniapps.common import upload_stokes, download_stokes
ni = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (130, 20, 17)
its = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = jr.miniapps.random_fields3d(ni, bcs="free_slip", iterMax=its - 1, nout=1000)
s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-30
h = _lib.default_handle()
outs = []
for variant in (1, 3):
    h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(variant))
    st, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, K, G, s.dt, None, kwargs=s.kwargs)
    outs.append(download_stokes(st))
for k in ("P", "txx", "txy", "txz", "tyz", "Vx", "Vy", "Vz"):
    a, b = outs[0][k], outs[1][k]
    m = checks.interior_mask3d(k, a.shape)
    d = (a != b) & m
    if d.any():
        idx = np.argwhere(d)
        print(k, "ndiff", len(idx), "i range", idx[:, 0].min(), idx[:, 0].max(), "unique i", np.unique(idx[:, 0])[:20], "j", np.unique(idx[:, 1])[:10], "k", np.unique(idx[:, 2])[:10],
              "maxabs", np.abs(a - b)[d].max())
    else:
        print(k, "identical")
