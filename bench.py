#!/usr/bin/env python3
"""bench.py -- PT-iterations/s of the 3D Stokes pseudo-transient loop (SolVi3D, 512^3 per GPU, fp64).

    python bench.py --gpus N --steps K --warmup W [--n 512]

One "step" = one PT iteration (stress sweep + velocity sweep + boundary conditions [+ halo exchange])
over one n^3 block per GPU, inputs resident in HBM.  N > 1: weak scaling, one process per GPU
(launched by torch.distributed.run), IGG-style block decomposition with RCCL halo exchange inside the
native library; torch.distributed (gloo) only carries the RCCL unique id, the barriers and the max-reduce.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

A_ALG = 360.0            # algorithmic bytes per cell per PT iteration (45 passes x 8 B)
A_STRESS = 28 * 8.0      # stress sweep: 21 reads + 7 writes
A_VELOCITY = 17 * 8.0    # velocity sweep: 14 reads + 3 writes
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# L2<->fabric bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950
# correction of MI355X_MICROARCH.md, + WRITE_SIZE; separate passes), collected offline on the same kernels at
# n = 512: profiles/r01_pmc_xcd_banded_traffic.txt, profiles/r01_pmc_fused_ylds_traffic.txt
#   k_stress3d_zb<512,1,4,xcd8>: FETCH_SIZE 13048845 KB, WRITE_SIZE 7410032 KB   (algorithmic: 21 + 7 passes of 1.074 GB)
#   k_fused3d<64,4,8,minw4,lowreg,xg1,shfl,ylds3,nt>: FETCH_SIZE 20010734 KB, WRITE_SIZE 11054315 KB  (needs 25 + 10 passes; fetched 38.2 + written 10.5;
#   profiles/r01_pmc_fused_final_traffic.txt; the 16-plane / 8-row-band form fetched 34.8 passes and was slower: part of the surplus is
#   served by the Infinity Cache, which FETCH_SIZE cannot tell from HBM)
PMC_TRAFFIC_STRESS_512 = (2 * 13048845.0 + 7410032.0) * 1024.0
PMC_TRAFFIC_FUSED_512 = (2 * 20010734.0 + 11054315.0) * 1024.0


def cpu_baseline(n_cpu: int, budget_s: float):
    """The oracle (CPU restatement, 6 unfused kernels, OpenMP) timed on this host's cores."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import numpy as np
    import oracle as orc
    from __graft_entry__ import load_package
    jr = load_package()
    from justrelax_jl_amd import checks
    import justrelax_jl_amd.grid as g
    g.finalize_global_grid()
    s = jr.miniapps.solvi3d(n_cpu)
    p = checks.oracle_params3d(orc, s)
    et = orc.compute_maxloc(s.arrays["eta"])
    orc.stokes3d_iteration(s.arrays, et, p)          # warm
    t0, it = time.perf_counter(), 0
    while True:
        orc.stokes3d_iteration(s.arrays, et, p)
        it += 1
        el = time.perf_counter() - t0
        if (el > budget_s and it >= 3) or it >= 2000:
            break
    cells_per_s = it * n_cpu ** 3 / el
    g.finalize_global_grid()
    return cells_per_s, it, el, orc.num_threads()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=512, help="local cells per dimension per GPU")
    ap.add_argument("--cpu-n", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="jrx_set_option kernel_variant (0 auto, 1 per-node, 2 z-marching sweeps, 3 fused wherever legal): tuning A/B only")
    ap.add_argument("--self-halo", nargs="?", const="xyz", default=None, metavar="DIMS",
                    help="diagnostic (1 GPU): IGG-periodic grid in DIMS (default xyz = all six faces) whose only neighbour is the rank "
                         "itself, planes routed through a one-rank RCCL communicator -- times the N > 1 code path (halo pack/send/recv/"
                         "unpack, shell fix-up) on one device")
    ap.add_argument("--dims", default="balanced", choices=["balanced", "yz"],
                    help="process grid for N > 1: balanced = IGG's default MPI_Dims_create factorisation ((2,2,2) for 8 GPUs, SURVEY 8e); "
                         "yz = (1, a, b) with x, the contiguous direction, never split (x faces are strided planes: their pack/unpack "
                         "and stress fix-up cost several times a y or z face) -- tuning option")
    args = ap.parse_args()

    # stdout carries exactly one JSON line (rank 0): native libraries that print banners on fd 1 (RCCL's version block
    # at communicator creation) are sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    jr = load_package()
    from justrelax_jl_amd import _lib, halo, stokes
    from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
    import justrelax_jl_amd.grid as grid

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (unique-id broadcast, barrier, max of the timings): gloo over loopback -- all ranks are on one node and the
        # container hostname may not resolve; the data path is the library's own RCCL communicator
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n = args.n
    self_halo = bool(args.self_halo) and world == 1
    if self_halo:
        os.environ["JRX_HALO_SELF_RCCL"] = "1"
        grid.init_global_grid(n, n, n, rank=0, nprocs=1, periodx=int("x" in args.self_halo), periody=int("y" in args.self_halo),
                              periodz=int("z" in args.self_halo))
    elif world > 1 and args.dims == "yz":
        dy = {2: 1, 4: 2, 8: 2, 16: 4}.get(world, 1)
        grid.init_global_grid(n, n, n, rank=rank, nprocs=world, dimx=1, dimy=dy, dimz=world // dy)
    else:
        grid.init_global_grid(n, n, n, rank=rank, nprocs=world)
    h = _lib.default_handle(local_rank)
    if args.variant:
        import ctypes as C
        h.call("jrx_set_option", C.c_char_p(b"kernel_variant"), C.c_int64(args.variant))
    if world > 1 or self_halo:
        halo.init_comm(h)
    uh = (lambda a: halo.update_halo_(a, ni=(n, n, n), handle=h)) if (world > 1 or self_halo) else None
    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend, update_halo=uh)
    jr.flow_bcs_(st, bcs, handle=h)
    if world > 1 or self_halo:
        halo.update_halo_(st.V.Vx, st.V.Vy, st.V.Vz, ni=(n, n, n), handle=h)
    ητ = jr.fzeros((n, n, n), dev)
    jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
    if world > 1 or self_halo:
        halo.update_halo_(ητ, ni=(n, n, n), handle=h)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
    if args.warmup > 0:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    tot_ms, sa_ms, sb_ms, sf_ms, sk_ms, _ = run_timed(stokes, st, pt, geo, bcs, ρg, K, G, ητ, dt, args.steps, h)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    barrier()
    if world > 1:
        t = torch.tensor([el, tot_ms, sa_ms, sb_ms, sf_ms, sk_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, tot_ms, sa_ms, sb_ms, sf_ms, sk_ms = t.tolist()

    if rank == 0:
        cells = float(n) ** 3
        fused = sf_ms > 0.0
        split = sa_ms > 0.0 and sb_ms > 0.0 and not fused       # N > 1: sweeps overlap with the halo exchange on two streams; price the whole iteration
        it_per_s = args.steps / el                       # PT iterations/s of the (global) problem
        value = world * it_per_s                         # n^3-block iterations/s summed over GPUs
        eff_gbs = A_ALG * cells * value / 1e9            # aggregate effective GB/s at 360 B/cell
        out = {
            "metric": f"PT-iterations/s (3D Stokes SolVi3D, {n}^3 fp64 block per GPU, block-iterations summed over GPUs)",
            "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SolVi3D {n}^3 per GPU (configs[{'3' if n == 512 else '2' if n == 256 else '?'}]): "
                                   "eta inclusion 1e-3, G=1, K=Inf, dt=Inf, free-slip, pure shear",
                       "local_grid": [n, n, n], "global_grid": [grid.nx_g(), grid.ny_g(), grid.nz_g()],
                       "decomposition": list(grid.global_grid().dims), "halo": "RCCL send/recv" if world > 1 else (f"diagnostic: periodic self-neighbour in {args.self_halo} through RCCL" if self_halo else "none")},
            "global_iterations_per_s": it_per_s,
            "effective_GBps_at_360B_per_cell": eff_gbs,
            "device_ms_per_step": tot_ms / args.steps,
            "roofline": None,
        }
        it_gbs = A_ALG * cells * (args.steps / (tot_ms * 1e-3)) / 1e9
        if fused:
            g = A_ALG * cells / (sk_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm",
                               "kernel": "k_fused3d: one PT iteration per launch (velocity sweep m + BCs + stress sweep m+1, ping-pong "
                                         "state); algorithmic 360 B/cell per launch (2-sweep floor of SURVEY 8d; the kernel itself needs "
                                         "25 reads + 10 writes = 280 B/cell)",
                               "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                               "traffic": PMC_TRAFFIC_FUSED_512 if n == 512 else None, "traffic_unit": "bytes per launch (PMC, offline)",
                               "algorithmic_bytes_per_launch": A_ALG * cells, "avg_launch_ms": sk_ms,
                               "launch_group_ms": sf_ms,
                               "whole_iteration": {"achieved": it_gbs, "frac": it_gbs / HBM_PEAK_GBS}}
        elif split:
            out["roofline"] = {"bound": "hbm",
                               "kernel": "stress sweep = k_stress3d_zb + 3 boundary-plane launches (21 array reads + 7 writes = 224 B/cell)",
                               "achieved": A_STRESS * cells / (sa_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": A_STRESS * cells / (sa_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "traffic": PMC_TRAFFIC_STRESS_512 if n == 512 else None, "traffic_unit": "bytes per launch (PMC, offline)",
                               "algorithmic_bytes_per_launch": A_STRESS * cells, "avg_launch_ms": sa_ms,
                               "velocity_sweep": {"achieved": A_VELOCITY * cells / (sb_ms * 1e-3) / 1e9, "avg_launch_ms": sb_ms},
                               "whole_iteration": {"achieved": it_gbs, "frac": it_gbs / HBM_PEAK_GBS}}
        else:
            out["roofline"] = {"bound": "hbm", "kernel": "whole PT iteration per GPU (360 B/cell; sweeps overlap the halo exchange)",
                               "achieved": it_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": it_gbs / HBM_PEAK_GBS, "traffic": None}
        if world == 1 and not args.no_cpu_baseline:
            del st, ρg, K, G, ητ
            cps, it, secs, thr = cpu_baseline(args.cpu_n, args.cpu_seconds)
            out["cpu_baseline"] = {"value": cps / cells, "unit": "it/s", "cores": thr, "kind": "port",
                                   "sample": f"oracle (6 unfused OpenMP kernels) on SolVi3D {args.cpu_n}^3, {it} iterations in "
                                             f"{secs:.1f} s, scaled by cell count to a {n}^3 block",
                                   "cell_updates_per_s": cps,
                                   "effective_GBps_at_600B_as_written": cps * 600.0 / 1e9}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_timed(stokes, st, pt, geo, bcs, ρg, K, G, ητ, dt, steps, h):
    return stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, steps, handle=h)


if __name__ == "__main__":
    main()
