#!/bin/bash
mkdir -p gpurun_out/r05t
bash scripts/clock_probe.sh 3 call_t | grep "clock probe" | cut -c1-160
python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_two_blocks.py -m gpu -q -x > gpurun_out/r05t/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r05t/tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/r05t/tests.log | head -8 | cut -c1-300
python - <<'PY' 2>&1 | tail -6
import sys
sys.path.insert(0, '.')
from __graft_entry__ import load_package
jr = load_package()
import torch
from justrelax_jl_amd import _lib, stokes
import justrelax_jl_amd.grid as grid
from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
for n, iters in ((512, 31), (256, 150)):
    h = _lib.default_handle(0)
    h.set_option("operand_cache", 1); h.set_option("viscous_limit", 0); h.set_option("zero_forces", 0)
    grid.finalize_global_grid(); grid.init_global_grid(n, n, n, rank=0, nprocs=1)
    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
    jr.flow_bcs_(st, bcs, handle=h)
    ητ = jr.fzeros((n, n, n), st.P.device); jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
    run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
    out = []
    for rep in range(3):
        for gh in (0, 4, 3):
            h.set_option("general_hif", gh)
            run(4)
            r = run(iters)
            out.append(f"gh{gh} kernel {r[4]:.3f} group {r[3]:.3f}")
    print(f"general form n {n}: " + " | ".join(out), flush=True)
    del st, ρg, K, G, ητ
    torch.cuda.empty_cache()
PY
