#!/usr/bin/env python3
"""In-process A/B of the fused kernel's tile shapes on ONE set of allocations (the placement lottery is per process): SolVi3D n^3, headline form, fused_tile 0 (64 x 4) / 3 (64 x 8), alternating."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
jr = load_package()
import torch  # noqa: E402
from justrelax_jl_amd import _lib, stokes  # noqa: E402
import justrelax_jl_amd.grid as grid  # noqa: E402
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 41
general = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # 1: the general form (viscous_limit = 0, zero_forces = 0): every operand and body force loaded
h = _lib.default_handle(0)
h.set_option("operand_cache", 1)
if general:
    h.set_option("viscous_limit", 0)
    h.set_option("zero_forces", 0)
grid.init_global_grid(n, n, n, rank=0, nprocs=1)
st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
jr.flow_bcs_(st, bcs, handle=h)
ητ = jr.fzeros((n, n, n), st.P.device)
jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
out = []
for rep in range(4):
    for tile in (0, 3):
        h.set_option("fused_tile", tile)
        run(5)
        out.append((tile, run(iters)[4]))
print(f"n {n} {'general form' if general else 'headline form'}: " + "  ".join(f"tile{t} {ms:.3f}" for t, ms in out), flush=True)
