"""debug: two-block inclusion run vs the oracle block by block"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
from __graft_entry__ import load_package
jr = load_package()
import oracle as orc
import test_gpu_two_blocks as T
import _blocks as B
from justrelax_jl_amd import _lib
L = _lib.load()
dims = tuple(int(c) for c in sys.argv[1])
nit = int(sys.argv[2])
n = (70, 13, 12)
kw = dict(iterMax=nit - 1, nout=8, verbose=False)
with T.TwoBlocks(n, dims) as tb:
    S = T._global_setup(jr, tb.ng, False, nit - 1, 8, seed=9)
    res, outs = T._solve_blocks(jr, tb, S, "split_sweeps", kw)
ng = tb.ng
b = S.flow_bcs
pl = orc.params3d(n, S.grid._di["center"], S.dt, dict(r=S.pt.r, theta_dtau=S.pt.θ_dτ, eta_dtau=S.pt.ηdτ, eps_rel=1e-30, eps_abs=1e-30),
                  iterMax=nit - 1, nout=8, free_slip=b.free_slip, no_slip=b.no_slip, periodic=b.periodic, ni_g=ng)
loc = [{k: B.local_block(v, n, ng, B.coords_of(tb.carts[r])) for k, v in S.arrays.items()} for r in range(2)]
et = [orc.compute_maxloc(l["eta"]) for l in loc]
B.exchange([[e] for e in et], n, tb.carts, L)
for it in range(nit):
    for r in range(2):
        orc.stokes3d_iteration(loc[r], et[r], pl)
    B.exchange([[l["Vx"], l["Vy"], l["Vz"]] for l in loc], n, tb.carts, L)
for r in range(2):
    for k in T.STATE + ("RP", "Rx", "Rz", "divV"):
        d = np.abs(outs[r][k] - loc[r][k])
        i = np.unravel_index(np.argmax(d), d.shape)
        print(r, k, d.max(), i, loc[r][k].shape)
