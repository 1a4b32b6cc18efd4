"""scripts/dbg_reroll.py -- contents of chunk-backed arrays across jrx_tuning_field_reroll (one array, all arrays, repeatedly): what the arrays hold afterwards and whether writes behind a re-mapping land; the record of the stale-translation finding in csrc/fieldpool.hip"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
jr = load_package()
import torch
from justrelax_jl_amd import _lib, arrays
h = _lib.Handle(0)
h.set_option("field_placement", 1); h.set_option("field_chunk_mib", 2)
arrays.use_library_arrays(h)
shapes = [(257, 130, 67), (1200, 1100), (300, 300, 30)]
rng = np.random.default_rng(3)
ts, ref = [], []
for sh in shapes:
    a = rng.standard_normal(sh); t = jr.fzeros(sh, "cuda"); t.copy_(torch.from_numpy(a).to("cuda")); ts.append(t); ref.append(a)
torch.cuda.synchronize()
def check(tag):
    for q, (t, a) in enumerate(zip(ts, ref)):
        b = t.cpu().numpy()
        bad = np.argwhere(b != a)
        print(tag, q, "mismatches", len(bad), "of", a.size, ("first flat index %d" % np.flatnonzero((b != a).ravel(order="F"))[0]) if len(bad) else "")
check("before")
def rewrite(tag, k):
    """write new values through the re-mapped ranges and read them back: do writes right behind a re-mapping land in the chunks later reads see?"""
    for t, a in zip(ts, ref):
        a += k; t.copy_(torch.from_numpy(a).to("cuda"))
    torch.cuda.synchronize(); check(tag)
h.call("jrx_tuning_field_reroll", C.c_void_p(ts[1].data_ptr())); torch.cuda.synchronize(); check("after one")
rewrite("first write after one", 1); rewrite("second write after one", 1)
h.call("jrx_tuning_field_reroll", C.c_void_p(0)); torch.cuda.synchronize()
rewrite("first write after all", 1); rewrite("second write after all", 1)
for r in range(5):
    h.call("jrx_tuning_field_reroll", C.c_void_p(0))
rewrite("first write after five more", 1); rewrite("second write after five more", 1)
