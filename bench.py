#!/usr/bin/env python3
"""bench.py -- PT-iterations/s of the 3D Stokes pseudo-transient loop (SolVi3D, 512^3 per GPU, fp64): the driver's contract line.

    python bench.py --gpus N --steps K --warmup W [--n 512] [--extras]

One "step" = one PT iteration (velocity sweep + boundary conditions + stress sweep [+ halo exchange]) over one n^3 block per GPU, inputs resident in HBM
(the boundary hands device pointers: there is no PCIe-inclusive variant of this path).  N > 1: weak scaling, one process per GPU, IGG-style block
decomposition, RCCL halo exchange inside the native library; torch.distributed (gloo) only carries the RCCL unique id, the barriers and the max-reduce.

stdout carries exactly ONE line: strict JSON (no NaN tokens), under 4 KB -- metric, value, ms_per_step, config, state_ok, roofline, cpu_baseline.  Everything
else (per-leg records, device clocks, the other configs with --extras) goes to bench_details.json (`details` in the line names it) and, abridged, to stderr.

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (the parent makes no GPU call); started by
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it finds WORLD_SIZE set and runs as one rank.

What is timed (N = 1), in this order, all on the library's DEFAULT options except where said:
  headline        W warm-up + K timed iterations of jrx_stokes3d_iterate_timed on arrays of the library's array constructor with option "field_placement" = 1
                  (--placement pool: every array ONE physical chunk picked at random from a pool spanning most of the free memory, mapped once, never moved;
                  --placement hipmalloc: torch's arrays).  dt = Inf (SolVi3D.jl:96), so the library runs the viscous-limit form of k_fused3d, priced at the
                  bytes THAT form has to move per SURVEY 8d's accounting (bench_extras.pricing); "operand_cache" stays at its default 0: the per-call operand
                  pass (3.6 ms at 512^3) is inside the timed region
  steady_state    100 more iterations of the same loop when K < 50
  general_form    the same batch with option viscous_limit = 0, zero_forces = 0: the kernel that loads every operand (any dt, any rho g), SURVEY 8d's 360 B/cell
  state_ok        the same W + K iterations from the same initial state on plain hipMalloc arrays: the int64 checksums of P, tau(6), V(3) must be equal
  cpu_baseline    the oracle's six-kernel OpenMP iteration on the host cores (bounded: 256^3 for ~8 s, scaled by cell count; --cpu-full-size on = at n^3)
"""
from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import bench_extras as X                                    # noqa: E402
from bench_extras import HBM_PEAK_GBS, pricing, fused_roofline, counters, nof_ran      # noqa: E402,F401  (re-exported: tests and scripts read them here)

LINE_LIMIT = 4096


def _clean(o):
    """strict JSON: NaN / Inf become null, floats keep 6 significant digits (the line must stay small)"""
    if isinstance(o, float):
        if o != o or o in (float("inf"), float("-inf")):
            return None
        return float(f"{o:.6g}")
    if isinstance(o, dict):
        return {k: _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    return o


def compact_line(full: dict) -> str:
    """the contract line: a fixed selection of `full`, strict JSON, < LINE_LIMIT bytes"""
    r = full.get("roofline") or {}
    g = r.get("general_form") or {}
    c = full.get("cpu_baseline") or {}
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "kernel_form", "local_grid", "global_grid", "decomposition", "halo", "arrays", "operand_cache") if k in cfg}
    for k in ("state_ok", "rccl_ranks", "degraded", "extras_incomplete"):
        if full.get(k) is not None:
            line[k] = full[k]
    if full.get("steady_state"):
        line["steady_state"] = {k: full["steady_state"].get(k) for k in ("steps", "value", "ms_per_step", "kernel_avg_launch_ms")}
    line["roofline"] = {k: r.get(k) for k in ("bound", "kernel_name", "form", "bytes_per_cell", "avg_launch_ms", "achieved", "peak", "unit", "frac", "traffic", "needed_bytes_per_launch",
                                              "frac_at_needed_bytes", "cells_per_launch", "launch_ms_per_rank", "avg_launch_ms_hipmalloc_arrays") if k in r}
    if g:
        line["roofline"]["general_form"] = {k: g.get(k) for k in ("bytes_per_cell", "avg_launch_ms", "frac", "frac_at_needed_bytes", "it_per_s", "traffic")}
    line["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "measured_at_n", "sample")} if c else None
    line["details"] = full.get("details")
    s = json.dumps(_clean(line), allow_nan=False, ensure_ascii=True, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:                                  # cannot happen with the fixed selection above unless a string grew: drop the free text first
        line["cpu_baseline"] = {k: v for k, v in (line["cpu_baseline"] or {}).items() if k != "sample"} or None
        line["config"].pop("halo", None)
        s = json.dumps(_clean(line), allow_nan=False, ensure_ascii=True, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def emit(json_fd, full: dict, path: str):
    """details file first, the abridged record on stderr, the contract line last and alone on stdout"""
    full["details"] = os.path.basename(path)
    try:
        Path(path).write_text(json.dumps(_clean(full), allow_nan=False, indent=1, ensure_ascii=False))
    except OSError as e:
        full["details"] = f"not written ({e})"
    brief = {k: v for k, v in full.items() if k in ("general_kernel", "hipmalloc_arrays", "steady_state", "device_state", "kernel_launch_counters", "placement", "solve_path")}
    sys.stderr.write("bench details: " + json.dumps(_clean(brief), allow_nan=False)[:6000] + "\n")
    os.write(json_fd, (compact_line(full) + "\n").encode())


def build_block(jr, h, n, dev, update_halo=None):
    """SolVi3D (miniapps/benchmarks/stokes3D/solvi/SolVi3D.jl:62-102) on the device + ητ: the argument tuple of stokes.iterate_timed_"""
    from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend, update_halo=update_halo)
    jr.flow_bcs_(st, bcs, handle=h)
    ητ = jr.fzeros((n, n, n), dev)
    jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
    return (st, pt, geo, bcs, ρg, K, G, ητ, dt)


def timed_batch(run, warm, steps, after_warm=None):
    import torch
    if warm > 0:
        run(warm)
    if after_warm:
        after_warm()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, res


def cpu_baseline_record(args, n):
    """bounded: the oracle at --cpu-n (256^3, ~8 s) scaled by cell count to an n^3 block; --cpu-full-size on/auto also measures at n^3"""
    cells = float(n) ** 3
    runs = []
    thr = 1
    for nc in args.cpu_n:
        ips, it, secs, thr = X.cpu_baseline(nc, args.cpu_seconds)
        runs.append({"n": nc, "it_per_s": ips, "iterations": it, "seconds": secs, "cell_updates_per_s": ips * nc ** 3, "effective_GBps_at_600B_as_written": ips * nc ** 3 * 600.0 / 1e9})
    full = None
    if n not in args.cpu_n and (args.cpu_full_size == "on" or (args.cpu_full_size == "auto" and X.mem_available_gb() >= 64.0 * (n / 512.0) ** 3)):
        try:
            ips, it, secs, thr = X.cpu_baseline(n, 0.0, min_iters=5)
            full = {"n": n, "it_per_s": ips, "iterations": it, "seconds": secs, "cell_updates_per_s": ips * n ** 3, "effective_GBps_at_600B_as_written": ips * n ** 3 * 600.0 / 1e9}
            runs.append(full)
        except MemoryError as e:
            runs.append({"n": n, "error": f"MemoryError: {e}"})
    if not runs:
        return None
    big = full or runs[-1]
    scaled = big["n"] != n
    return {"value": big["cell_updates_per_s"] / cells, "unit": "it/s", "cores": thr, "host_logical_cpus": os.cpu_count(), "kind": "port", "measured_at_n": big["n"],
            "sample": (f"oracle (6 unfused OpenMP kernels, {thr} threads) on SolVi3D {big['n']}^3: {big['iterations']} iterations in {big['seconds']:.1f} s = {big['it_per_s']:.3f} it/s"
                       + (f", scaled by cell count to a {n}^3 block" if scaled else ", measured at the metric's size")),
            "measured": runs}


def run_single(args, json_fd) -> int:
    import ctypes as C
    import torch
    from __graft_entry__ import load_package
    jr = load_package()
    from justrelax_jl_amd import _lib, halo, stokes
    import justrelax_jl_amd.grid as grid

    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py: no GPU visible")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n = args.n
    self_halo = bool(args.self_halo)
    if self_halo:
        grid.init_global_grid(n, n, n, rank=0, nprocs=1, periodx=int("x" in args.self_halo), periody=int("y" in args.self_halo), periodz=int("z" in args.self_halo))
    else:
        grid.init_global_grid(n, n, n, rank=0, nprocs=1)
    h = _lib.default_handle(0)
    if args.variant:
        h.set_option("kernel_variant", args.variant)
    for kv in args.option:
        k, v = kv.split("=")
        h.set_option(k, int(v))
    rccl_ranks = 0
    uh = None
    if self_halo:
        halo.init_comm(h, self_rccl=True)
        cnt = C.c_int32(0)
        h.call("jrx_comm_count", C.byref(cnt))
        rccl_ranks = cnt.value
        uh = lambda a: halo.update_halo_(a, ni=(n, n, n), handle=h)

    def make(pool):
        pa = X.PoolArrays(h, n, pool)
        with pa:
            blk = build_block(jr, h, n, dev, uh)
        if self_halo:
            halo.update_halo_(blk[0].V.Vx, blk[0].V.Vy, blk[0].V.Vz, blk[7], ni=(n, n, n), handle=h)
        return pa, blk

    pool, placement_note = args.placement == "pool", None
    t_setup = time.perf_counter()
    try:
        pa, blk = make(pool)
    except _lib.JrxError as e:              # the driver refuses the virtual-memory-management calls: torch's arrays, and the line says so
        if not pool:
            raise
        placement_note = f"library arrays refused ({e}); torch's arrays"
        from justrelax_jl_amd import arrays as _arrays
        _arrays.use_library_arrays(None)
        h.set_option("field_placement", 0)
        pool = False
        pa, blk = make(False)
    setup_s = time.perf_counter() - t_setup
    run = lambda k: stokes.iterate_timed_(*blk, k, handle=h)
    dt = blk[8]

    try:
        pp = torch.cuda.get_device_properties(0)
        pci = f"{pp.pci_bus_id:02x}:{pp.pci_device_id:02x}.0"
    except Exception:       # noqa: BLE001
        pci = None
    dstate = X.DeviceState(pci)
    f0 = [None]

    def after_warm():
        if args.warmup > 0:
            pa.trim()                      # the library's second state set exists now (first driver call): the pool's unused chunks go back to the driver
        dstate.start()
        f0[0] = counters(h)
    el, (tot_ms, sa_ms, sb_ms, sf_ms, sk_ms, kcells) = timed_batch(run, args.warmup, args.steps, after_warm)
    dev_state = dstate.stop()
    pa.trim()
    pr = pricing(h, dt, nof_ran(h, f0[0])) if f0[0] else pricing(h, dt, 0)
    sums_a = X.state_checksums(blk[0]) if not args.no_state_check else None
    pool_stats = None
    if pool:
        s6 = (C.c_int64 * 6)()
        h.call("jrx_field_stats", s6)
        pool_stats = {"live_arrays": s6[0], "live_bytes": s6[1], "chunks_created": s6[2], "hipMemCreate_s": s6[4] / 1e6, "map_s": s6[5] / 1e6}

    cells = float(n) ** 3
    fused = sf_ms > 0.0
    value = args.steps / el
    full = {
        "metric": f"PT-iterations/s (3D Stokes SolVi3D, {n}^3 fp64 block per GPU, block-iterations summed over GPUs)",
        "value": value, "unit": "it/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SolVi3D {n}^3 per GPU (configs[{'3' if n == 512 else '2' if n == 256 else '?'}]): eta inclusion 1e-3, G=1, K=Inf, dt=Inf, free-slip, pure shear",
                   "kernel_form": pr["form"], "local_grid": [n, n, n], "global_grid": [grid.nx_g(), grid.ny_g(), grid.nz_g()], "decomposition": list(grid.global_grid().dims),
                   "halo": f"diagnostic: periodic self-neighbour in {args.self_halo} through RCCL" if self_halo else "none",
                   "arrays": "library constructor, field_placement=1 (pool-dealt chunks, mapped once)" if pool else "torch (hipMalloc)", "operand_cache": h.get_option("operand_cache")},
        "rccl_ranks": rccl_ranks or None, "placement": {"note": placement_note, "pool": pool_stats, "setup_s": setup_s},
        "effective_GBps": pr["alg"] * cells * value / 1e9, "device_ms_per_step": tot_ms / args.steps, "device_state": dev_state, "roofline": None,
    }
    it_gbs = pr["alg"] * cells * (args.steps / (tot_ms * 1e-3)) / 1e9
    if fused:
        full["roofline"] = fused_roofline(pr, n, sk_ms, sf_ms, kcells, it_gbs)
        full["roofline"]["kernel_name"] = "k_fused3d"
        if not self_halo and not args.variant:
            X.check_priced_kernel(h, pr, f0[0], full)
    else:
        full["roofline"] = {"bound": "hbm", "kernel_name": "whole PT iteration (un-fused sweeps)", "kernel": f"whole PT iteration per GPU ({pr['alg']:.0f} B/cell, form: {pr['form']})", "form": pr["form"],
                            "bytes_per_cell": pr["alg"], "avg_launch_ms": tot_ms / args.steps, "achieved": it_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": it_gbs / HBM_PEAK_GBS, "traffic": None}
    full["roofline"]["launch_ms_per_rank"] = [sk_ms if fused else None]

    # ---- steady state: a short requested batch (the driver times 20 steps = 0.1 s) also holds the per-call operand pass; 100 steps of the same loop beside it
    if args.steps < 50 and not args.no_steady_state:
        sel, sres = timed_batch(run, 0, 100)
        full["steady_state"] = {"steps": 100, "value": 100 / sel, "ms_per_step": sel / 100 * 1e3, "kernel_avg_launch_ms": sres[4] if sres[4] > 0 else None}

    # ---- the general form of the same kernel (any dt, every operand and body-force array loaded): SURVEY 8d's 360 B/cell, same arrays, same batch length
    if fused and pr["form"].startswith("viscous_limit") and not args.no_general_kernel and not self_halo:
        try:
            h.set_option("viscous_limit", 0)
            h.set_option("zero_forces", 0)
            fg = [None]
            gel, gres = timed_batch(run, max(args.warmup, 2), args.steps, lambda: fg.__setitem__(0, counters(h)))
            prg = pricing(h, dt, nof_ran(h, fg[0]))
            gr = fused_roofline(prg, n, gres[4], gres[3], gres[5], prg["alg"] * cells * (args.steps / (gres[0] * 1e-3)) / 1e9)
            full["general_kernel"] = {"it_per_s": args.steps / gel, "ms_per_step": gel / args.steps * 1e3, "steps": args.steps, "roofline": gr}
            full["roofline"]["general_form"] = {"bytes_per_cell": gr["bytes_per_cell"], "avg_launch_ms": gr["avg_launch_ms"], "achieved": gr["achieved"], "frac": gr["frac"],
                                                "frac_at_needed_bytes": gr.get("frac_at_needed_bytes"), "it_per_s": args.steps / gel, "traffic": gr.get("traffic")}
        except Exception as e:      # noqa: BLE001 -- a side leg must not lose the headline
            full["general_kernel"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            h.set_option("viscous_limit", 1)
            h.set_option("zero_forces", 1)

    if args.extras and not self_halo:
        try:
            full["solve_path"] = X.solve_path(jr, h, blk[0], blk[1], blk[2], blk[3], blk[4], blk[5], blk[6], dt, args.solve_iters, n, pr)
        except Exception as e:      # noqa: BLE001
            full["solve_path"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- state_ok: the same W + K iterations from the same initial state on plain hipMalloc arrays must leave the same bits in P, tau, V
    rc = 0
    blk = run = None
    pa.done()
    if sums_a is not None and not self_halo:
        try:
            pb, blk = make(False)
            run = lambda k: stokes.iterate_timed_(*blk, k, handle=h)
            bel, bres = timed_batch(run, args.warmup, args.steps)
            sums_b = X.state_checksums(blk[0])
            full["state_ok"] = sums_a == sums_b
            full["hipmalloc_arrays"] = {"it_per_s": args.steps / bel, "kernel_avg_launch_ms": bres[4] if bres[4] > 0 else None,
                                        "what": "the same W + K iterations on torch's arrays (plain hipMalloc) in the same process; its checksums are what state_ok compares with"}
            full["roofline"]["avg_launch_ms_hipmalloc_arrays"] = bres[4] if bres[4] > 0 else None
            if not full["state_ok"]:
                full["state_checksums"] = {"headline_arrays": sums_a, "hipmalloc_arrays": sums_b}
                sys.stderr.write("bench.py: STATE MISMATCH between the headline's arrays and hipMalloc arrays after the same iterations\n")
                rc = 1
        except Exception as e:      # noqa: BLE001
            full["state_ok"] = None
            full["hipmalloc_arrays"] = {"error": f"{type(e).__name__}: {e}"}
        blk = run = None
        torch.cuda.empty_cache()

    if args.extras and not self_halo:
        full["other_configs"] = X.other_configs(jr, h)
    if not args.no_cpu_baseline:
        grid.finalize_global_grid()
        full["cpu_baseline"] = cpu_baseline_record(args, n)
    emit(json_fd, full, args.details)
    return rc


def run_multi(args, world, rank, local_rank, json_fd) -> int:
    """N > 1 (one process per GPU).  Headline = the requested K steps on the RCCL transport with IGG's balanced decomposition (north_star's configuration).  With --extras, behind
    it in the same processes: the chain breakdown per rank, the ipc transport, one process driving all devices (local_peer), the other decomposition (bench_extras)."""
    from datetime import timedelta
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # control plane only (ids, barriers, max of the timings, gathers): gloo over loopback; the data path is the library's own transport
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=timedelta(minutes=60))
    ctl = X.Control(rank, world)
    n = args.n
    if args.dry_transports:
        R, local_peer = X.DryRanks(args, rank, world), X.dry_local_peer
        pr = {"form": "dry", "alg": X.A_ALG_VISC, "needed": X.A_NEEDED_VISC, "pmc": None, "pmc_source": None, "kernel": "dry run"}
    else:
        import torch
        from __graft_entry__ import load_package
        jr = load_package()
        if args.same_device:
            local_rank = 0
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} needs device {local_rank}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local_rank)
        R = X.GpuRanks(args, jr, rank, world, local_rank)
        if args.variant:
            R.h.set_option("kernel_variant", args.variant)
        for kv in args.option:
            k, v = kv.split("=")
            R.h.set_option(k, int(v))
        local_peer = lambda w, mode, steps, warm, n1: X.local_peer_leg(jr, args, w, mode, steps, warm, n1)
    dims = R.build(args.dims)
    T0 = args.default_transport
    R.connect(T0)
    ranks = R.comm_count()
    if ranks != world:
        raise SystemExit(f"bench.py: the {T0} communicator has {ranks} ranks, expected {world}")
    if args.warmup > 0:
        R.run(args.warmup)
    R.sync(); ctl.barrier()
    f0 = None if args.dry_transports else counters(R.h)
    t0 = time.perf_counter()
    tot_ms, sa_ms, sb_ms, sf_ms, sk_ms, kcells = R.run(args.steps)
    R.sync()
    el = time.perf_counter() - t0
    if not args.dry_transports:
        pr = pricing(R.h, R.blk[8], nof_ran(R.h, f0))
    ctl.barrier()
    # per rank: the launch time of the dominant kernel and where the arrays came from -- the slowest rank paces a weak-scaling run, and it should be visible which one it was
    per_rank = ctl.gather({"rank": rank, "k_fused3d_ms": sk_ms, "arrays": getattr(R, "placement", None)})
    el, tot_ms, sa_ms, sb_ms, sf_ms, sk_ms, kcells = ctl.max([el, tot_ms, sa_ms, sb_ms, sf_ms, sk_ms, kcells])
    steady = None
    if args.steps < 50 and not args.no_steady_state:
        ctl.barrier()
        t1 = time.perf_counter()
        sres = R.run(100)
        R.sync()
        sel = ctl.max([time.perf_counter() - t1])[0]
        ctl.barrier()
        steady = {"steps": 100, "value": world * 100 / sel, "ms_per_step": sel / 100 * 1e3, "kernel_avg_launch_ms": sres[4] if sres[4] > 0 else None}
    cells = float(n) ** 3
    value = world * args.steps / el
    full = {
        "metric": f"PT-iterations/s (3D Stokes SolVi3D, {n}^3 fp64 block per GPU, block-iterations summed over GPUs)",
        "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "steady_state": steady,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SolVi3D {n}^3 per GPU (configs[3]: 512^3 per GPU, weak scaling): eta inclusion 1e-3, G=1, K=Inf, dt=Inf, free-slip, pure shear",
                   "kernel_form": pr["form"], "local_grid": [n, n, n], "global_grid": [d * (n - 2) + 2 for d in dims], "decomposition": list(dims),
                   "halo": {"rccl": "RCCL grouped send/recv per dimension", "ipc": "copy engines between processes (IPC memory handles)"}[T0] + (" [diagnostic: every rank on device 0]" if args.same_device else ""),
                   "arrays": (per_rank[0] or {}).get("arrays")},
        "rccl_ranks": ranks if T0 == "rccl" else 0, "default_transport": T0, "global_iterations_per_s": args.steps / el,
        "effective_GBps": pr["alg"] * cells * value / 1e9, "device_ms_per_step": tot_ms / args.steps, "roofline": None,
        "transports": {T0: {"ranks": ranks, "steps": args.steps, "it_per_s": value, "ms_per_step": el / args.steps * 1e3, "decomposition": list(dims), "pipeline": R.pipeline()}},
    }
    it_gbs = pr["alg"] * cells * (args.steps / (tot_ms * 1e-3)) / 1e9
    if sf_ms > 0.0:
        full["roofline"] = fused_roofline(pr, n, sk_ms, sf_ms, kcells, it_gbs)
        full["roofline"]["kernel_name"] = "k_fused3d"
    else:
        full["roofline"] = {"bound": "hbm", "kernel_name": "whole PT iteration", "kernel": f"whole PT iteration per GPU ({pr['alg']:.0f} B/cell, form: {pr['form']}; sweeps overlap the halo exchange)",
                            "form": pr["form"], "bytes_per_cell": pr["alg"], "avg_launch_ms": tot_ms / args.steps, "achieved": it_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": it_gbs / HBM_PEAK_GBS, "traffic": None}
    if not args.dry_transports:
        X.check_priced_kernel(R.h, pr, f0, full)
    full["roofline"]["launch_ms_per_rank"] = [p.get("k_fused3d_ms") for p in per_rank]

    def out(note=None):
        if note:
            full["extras_incomplete"] = note
            full["degraded"] = True
        emit(json_fd, full, args.details)
    if not args.extras:
        if rank == 0:
            out()
    else:
        wd = X.Watchdog(args.extras_budget, rank, out)
        try:
            X.multi_rank_extras(R, ctl, args, full, local_peer, wd)
            note = None
        except Exception as e:      # noqa: BLE001 -- the headline is kept
            note = f"{type(e).__name__}: {e}"
        wd.cancel()
        if rank == 0:
            out(note)
    R.disconnect()
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:       # noqa: BLE001 -- a rank that left early must not turn a finished measurement into a failure
        pass
    return 0


def run_rank(args) -> int:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_launch:
        if rank == 0:
            print(json.dumps({"dry_launch": True, "world": world, "rank": rank, "local_rank": local_rank, "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}), flush=True)
        else:
            sys.stderr.write(f"dry-launch rank {rank}/{world} local_rank {local_rank}\n")
        marker = os.environ.get("JRX_DRY_LAUNCH_DIR")
        if marker:
            Path(marker, f"rank{rank}.json").write_text(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        return 0
    # stdout carries exactly one JSON line (rank 0): native libraries that print banners on fd 1 (RCCL's version block at communicator creation) are sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        return run_multi(args, world, rank, local_rank, json_fd)
    return run_single(args, json_fd)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = X.parse_args(argv)
    if args.ipc_helper:
        return X.ipc_helper(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return X.launch_ranks(args, argv)
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.extras and not (args.dry_launch or args.dry_transports or args.self_halo):
        X.start_ipc_helpers(args)             # the two-process leg of --extras: its rank processes must exist before this process touches the GPU
    try:
        return run_rank(args)
    finally:
        X.stop_ipc_helpers()


if __name__ == "__main__":
    sys.exit(main())
