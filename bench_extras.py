"""bench_extras.py -- everything of the benchmark that is NOT the driver's contract line (bench.py): the pricing of the kernel forms, the CPU baseline, the other BASELINE configs
(SolVi3D 256^3, SolCx 512^2, shear band 1024^2, thermal diffusion 256^2) and the 3D VEP / 3D thermal paths at 256^3, the coupled-blocks legs on one device (in-process and
two-process transports), and for N > 1 the transports x decompositions behind the headline.  bench.py imports it; `python bench.py --extras` runs these legs and writes them to
bench_details.json.  Nothing here is printed on stdout."""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
SCRIPT = ROOT / "bench.py"          # what the rank / helper processes are started from
sys.path.insert(0, str(ROOT))

A_ALG = 360.0            # algorithmic bytes per cell per PT iteration (45 passes x 8 B)
A_NEEDED_FUSED = 35 * 8.0   # what k_fused3d itself has to move: 25 array reads + 10 writes = 280 B/cell (V handed from the velocity to the stress phase in LDS)
# dt = Inf (SolVi3D, the headline workload): 1/(G dt) = 1/(K dt) = 1/dt = 0 exactly, and the ten operand arrays they multiply (six old stresses, P0, K, G, Q)
# cannot change any result; the library's default kernel for that limit does not load them (option viscous_limit).  The two sweeps of SURVEY 8d without
# those ten arrays are 35 passes -- the figure that launch is priced at (pricing it at 360 B/cell would credit it with bytes nobody has to move).
A_ALG_VISC = 35 * 8.0
A_NEEDED_VISC = 25 * 8.0    # what the viscous-limit k_fused3d itself has to move: 15 array reads + 10 writes = 200 B/cell
# SolVi3D hands three body-force arrays of zeros (SolVi3D.jl:102).  The one-launch viscous-limit kernel does not load ρg arrays in which the operand pass of the driver call has
# found nothing but +0.0 (x - 0.5 (0 + 0) = x for every x; tuning switch zero_forces): a launch of that form is priced WITHOUT those passes -- 8 B/cell less per array, 32 passes =
# 256 B/cell for SolVi3D -- and the same kernel with the loads (280 B/cell) is timed beside it as `with_body_forces`.
FORMS_NOF = {1: ("viscous_limit_gravity_along_z", 2), 2: ("viscous_limit_no_body_forces", 3)}
A_STRESS = 28 * 8.0      # stress sweep: 21 reads + 7 writes
A_VELOCITY = 17 * 8.0    # velocity sweep: 14 reads + 3 writes
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
A_STRESS_VISC = 18 * 8.0  # ... in the viscous limit: 11 reads + 7 writes (no tau_o, P0, K, G, Q)


class DeviceState:
    """sclk / power / temperatures of this rank's device (hwmon files under /sys/class/drm/card*/device/hwmon), sampled by a thread while a batch runs.  Why it is in the line: the fused kernel
    draws ~1.36 kW of the 1.4 kW cap, so its rate follows the clock the power management sustains (2.30 - 2.40 GHz on the boxes seen) -- the spread between boxes, processes and ranks that
    rounds 3-4 took for a memory-placement lottery (profiles/r05_placement.txt: new physical chunks under fixed addresses, new virtual layouts, other streams, other code copies change nothing)."""

    def __init__(self, pci_bus_id=None):
        import glob
        self.paths = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            hw = sorted(glob.glob(d + "/hwmon/hwmon*"))
            if hw and os.path.exists(hw[0] + "/freq1_input"):
                try:
                    bus = os.path.basename(os.path.realpath(d))
                except OSError:
                    bus = ""
                self.paths.append((bus, hw[0]))
        self.want = (pci_bus_id or "").lower()
        self.samples, self._stop, self._th = [], None, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return float("nan")

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append([(self._read(h + "/freq1_input"), self._read(h + "/power1_input"), self._read(h + "/temp2_input"), self._read(h + "/temp3_input")) for _, h in self.paths])
            self._stop.wait(0.02)

    def start(self):
        import threading
        self.samples, self._stop = [], threading.Event()
        if self.paths:
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def stop(self):
        if self._th:
            self._stop.set()
            self._th.join()
            self._th = None
        if not self.samples:
            return None
        nd = len(self.paths)
        idx = [i for i, (bus, _) in enumerate(self.paths) if self.want and bus.lower().endswith(self.want[-7:])]
        peak = [max(smp[i][1] for smp in self.samples) for i in range(nd)]
        me = idx[0] if idx else max(range(nd), key=lambda i: peak[i])         # by PCI address when it can be matched, else the device that drew the most
        busy = [smp[me] for smp in self.samples if smp[me][1] >= 0.6 * peak[me]] or [smp[me] for smp in self.samples]
        med = lambda v: sorted(v)[len(v) // 2]
        cap = self._read(self.paths[me][1] + "/power1_cap")
        return {"sclk_mhz": med([b[0] for b in busy]) / 1e6, "power_w": med([b[1] for b in busy]) / 1e6, "power_cap_w": cap / 1e6 if cap == cap else None,
                "junction_c": med([b[2] for b in busy]) / 1e3, "hbm_c": med([b[3] for b in busy]) / 1e3, "samples": len(busy), "matched_by": "pci" if idx else "highest power",
                "devices_visible": nd, "other_devices_above_600w": sum(1 for i in range(nd) if i != me and peak[i] > 600e6)}


def load_pmc():
    """L2<->fabric bytes per launch of the dominant kernels from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE;
    separate passes), collected offline at n = 512 and kept in profiles/pmc_traffic.json together with the sha256 of csrc/stokes3d_kernels.hpp and the git commit they
    were taken at (scripts/pmc_traffic.py writes the file).  A kernel source that has changed since makes the figures stale: `traffic` is then null."""
    import hashlib
    empty = {"k_fused3d_general": None, "k_fused3d_visc": None, "k_stress3d_zb_general": None, "k_stress3d_zb_visc": None, "source": None, "stale": True}
    try:
        d = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        have = hashlib.sha256((ROOT / "justrelax.jl_amd" / "csrc" / "stokes3d_kernels.hpp").read_bytes()).hexdigest()
    except (OSError, ValueError):
        return empty
    if d.get("kernels_sha256") != have:
        return dict(empty, source=f"{d.get('source')} @ {d.get('git_head')}: STALE, csrc/stokes3d_kernels.hpp has changed since")
    return dict(d, source=f"{d.get('source')} @ {d.get('git_head')}", stale=False)


PMC = load_pmc()
PMC_TRAFFIC_STRESS_512 = PMC["k_stress3d_zb_general"]          # k_stress3d_zb, general form
PMC_TRAFFIC_STRESS_VISC_512 = PMC["k_stress3d_zb_visc"]        # k_stress3d_zb, viscous-limit form
PMC_TRAFFIC_FUSED_512 = PMC["k_fused3d_general"]               # k_fused3d, general form (38.2 + 10.5 array passes in round 3)
PMC_TRAFFIC_VISC_512 = PMC["k_fused3d_visc"]                   # k_fused3d, viscous-limit form (26.6 + 10.6 array passes in round 3)
PMC_SOURCE = {"stress": PMC["source"], "fused": PMC["source"], "visc": PMC["source"]}


NOF_COUNTERS = ("stat_fused3d", "stat_fused3d_visc", "stat_fused3d_nof1", "stat_fused3d_nof2")


def counters(h):
    return [h.get_option(k) for k in NOF_COUNTERS]


def nof_ran(h, before):
    """which body-force arrays the fused launches since `before` (= counters(h)) did not load: 0 none, 1 ρg_x and ρg_y, 2 all three -- read from the library's launch
    counters, so that a launch is never priced at a form that did not run"""
    d = [b - a for a, b in zip(before, counters(h))]
    for lvl in (1, 2):
        if d[0] > 0 and d[1 + lvl] == d[0]:
            return lvl
    if d[2] or d[3]:
        raise SystemExit(f"bench.py: the fused launches of one batch ran in different forms {dict(zip(NOF_COUNTERS, d))}")
    return 0


def pricing(h, dt, nof=0):
    """what one launch of the fused kernel is priced at: the form of the kernel that runs (h: handle, dt: the time step handed to the solver, nof: nof_ran() of the batch)"""
    import math
    if math.isinf(dt) and h.get_option("viscous_limit") == 1 and h.get_option("fused_ylds") == 1 and nof:
        form, na = FORMS_NOF[nof]
        return {"form": form, "alg": A_ALG_VISC - 8.0 * na, "needed": A_NEEDED_VISC - 8.0 * na, "pmc": PMC.get(f"k_fused3d_visc_nof{nof}"), "pmc_source": PMC_SOURCE["visc"], "nof": nof,
                "kernel": f"k_fused3d<...,VISC=1,HIF=1,NOF={nof}>: one PT iteration per launch (velocity sweep m + BCs + stress sweep m+1 + the high-face node layers, ping-pong state) in the "
                          f"viscous limit dt = Inf of the workload, whose {'three body-force arrays are' if nof == 2 else 'body-force arrays ρg_x, ρg_y are'} +0.0 in every entry (SolVi3D.jl:102; "
                          f"the operand pass of the driver call checks the bits): algorithmic {A_ALG_VISC - 8.0 * na:.0f} B/cell per launch = SURVEY 8d's two sweeps (45 passes) less the ten operand "
                          f"arrays whose factor 1/(G dt), 1/(K dt), 1/dt is exactly 0 and less the {na} zero body-force arrays, none of which this form loads; the kernel itself needs "
                          f"{15 - na} reads + 10 writes.  The same kernel with the body-force loads (280 B/cell) is the `with_body_forces` entry of this line, the general form (360 B/cell) `general_kernel`"}
    if math.isinf(dt) and h.get_option("viscous_limit") == 1 and h.get_option("fused_ylds") == 1:
        return {"form": "viscous_limit", "nof": 0, "alg": A_ALG_VISC, "needed": A_NEEDED_VISC, "pmc": PMC_TRAFFIC_VISC_512, "pmc_source": PMC_SOURCE["visc"],
                "kernel": "k_fused3d<...,VISC=1,TAG=0>: one PT iteration per launch (velocity sweep m + BCs + stress sweep m+1, ping-pong state) in the viscous limit dt = Inf of "
                          "the workload: algorithmic 280 B/cell per launch = SURVEY 8d's two sweeps (45 passes) less the ten operand arrays whose factor 1/(G dt), 1/(K dt), 1/dt "
                          "is exactly 0 (old stresses, P0, K, G, Q), which this form does not load; the kernel itself needs 15 reads + 10 writes = 200 B/cell.  The general form "
                          "priced at 360 B/cell is the `general_kernel` entry of this line"}
    if nof:
        form, na = FORMS_NOF[nof]
        return {"form": form.replace("viscous_limit", "general"), "nof": nof, "alg": A_ALG - 8.0 * na, "needed": A_NEEDED_FUSED - 8.0 * na, "pmc": PMC.get(f"k_fused3d_general_nof{nof}"), "pmc_source": PMC_SOURCE["fused"],
                "kernel": f"k_fused3d<...,NOF={nof}>: the general form (any dt) of the fused iteration for a workload whose {'three body-force arrays are' if nof == 2 else 'body-force arrays ρg_x, ρg_y are'} +0.0 in "
                          f"every entry and are not loaded: algorithmic {A_ALG - 8.0 * na:.0f} B/cell per launch = SURVEY 8d's 45 passes less those {na}; the kernel itself needs {25 - na} reads + 10 writes"}
    return {"form": "general", "nof": 0, "alg": A_ALG, "needed": A_NEEDED_FUSED, "pmc": PMC_TRAFFIC_FUSED_512, "pmc_source": PMC_SOURCE["fused"],
            "kernel": "k_fused3d<...,TAG=0>: one PT iteration per launch (velocity sweep m + BCs + stress sweep m+1, ping-pong "
                      "state); algorithmic 360 B/cell per launch (2-sweep floor of SURVEY 8d; the kernel itself needs "
                      "25 reads + 10 writes = 280 B/cell)"}


def fused_roofline(pr, n, sk_ms, sf_ms, kcells, it_gbs=None):
    cells = float(n) ** 3
    kcells = kcells or cells
    g = pr["alg"] * kcells / (sk_ms * 1e-3) / 1e9
    whole = kcells == cells
    tr = (pr["pmc"] * kcells / cells) if (n == 512 and pr["pmc"]) else None
    out = {"bound": "hbm",
           "kernel": pr["kernel"] + ("" if whole else
                     f"; this launch covers the {kcells:.0f} cells of the tiles that touch no high face ({kcells / cells:.4f} of the block), "
                     "the high-face tiles (k_fused3d<...,TAG=1>) and the boundary stress layers run beside it on a second stream"),
           "form": pr["form"], "bytes_per_cell": pr["alg"],
           "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
           "traffic": tr,
           "traffic_unit": "bytes per launch (PMC, offline, whole-block launch scaled by the cell share of this launch)",
           "traffic_source": pr["pmc_source"],
           "traffic_ratio": tr / (pr["alg"] * kcells) if tr else None,
           "needed_bytes_per_launch": pr["needed"] * kcells,
           "frac_at_needed_bytes": pr["needed"] * kcells / (sk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,      # every array the kernel reads or writes counted once: the compulsory traffic of this fused form
           "traffic_over_needed": tr / (pr["needed"] * kcells) if tr else None,
           "cells_per_launch": kcells,
           "algorithmic_bytes_per_launch": pr["alg"] * kcells, "avg_launch_ms": sk_ms,
           "launch_group_ms": sf_ms}
    if it_gbs is not None:
        out["whole_iteration"] = {"achieved": it_gbs, "frac": it_gbs / HBM_PEAK_GBS}
    return out


# ------------------------------------------------------------------------------------------------ launching
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=512, help="local cells per dimension per GPU")
    ap.add_argument("--cpu-n", type=int, nargs="*", default=[256], help="oracle sizes of the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU budget per oracle size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="accepted for compatibility: the extra legs are off unless --extras is given")
    ap.add_argument("--no-general-kernel", action="store_true", help="skip the leg that times the general form of k_fused3d (option viscous_limit = 0) beside the headline")
    ap.add_argument("--placement", choices=["pool", "hipmalloc"], default="pool",
                    help="where the arrays of the headline come from: pool = the library's array constructor with option field_placement = 1 (every array ONE physical chunk picked at random "
                         "from a pool that spans most of the free memory, mapped once at a fresh virtual range: csrc/fieldpool.hip); hipmalloc = torch's arrays (plain hipMalloc)")
    ap.add_argument("--extras", action="store_true", help="also run the legs of bench_extras.py (other configs, coupled blocks, solve path; N > 1: transports and decompositions) and write them to --details")
    ap.add_argument("--details", default=str(ROOT / "bench_details.json"), help="where the full record goes (the stdout line stays compact)")
    ap.add_argument("--no-state-check", action="store_true", help="skip the state_ok leg (the same batch on hipMalloc arrays, checksums compared)")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the extra 100-step batch that follows a requested batch of fewer than 50 steps")
    ap.add_argument("--cpu-full-size", choices=["auto", "on", "off"], default="off",
                    help="cpu_baseline also measured at the metric's own size (n^3, 5 iterations; ~1 minute of host work at 512^3): auto = when MemAvailable >= 64 GB")
    ap.add_argument("--solve-iters", type=int, default=399, help="iterMax of the solve_path leg (nout = 100)")
    ap.add_argument("--variant", type=int, default=0, help="jrx_set_option kernel_variant (0 auto, 1 per-node, 2 z-marching sweeps, 3 fused wherever legal): tuning A/B only")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT", help="jrx_set_option(KEY, INT) before the run (tuning A/B)")
    ap.add_argument("--self-halo", nargs="?", const="xyz", default=None, metavar="DIMS",
                    help="diagnostic (1 GPU): IGG-periodic grid in DIMS (default xyz = all six faces) whose only neighbour is the rank "
                         "itself, planes routed through a one-rank RCCL communicator -- times the N > 1 code path (halo pack/send/recv/"
                         "unpack, shell fix-up) on one device")
    ap.add_argument("--dims", default="balanced", choices=["balanced", "yz"],
                    help="process grid for N > 1: balanced = IGG's default MPI_Dims_create factorisation ((2,2,2) for 8 GPUs, SURVEY 8e); "
                         "yz = (1, a, b) with x, the contiguous direction, never split -- tuning option")
    ap.add_argument("--leg-steps", type=int, default=60, help="N > 1: steps of every leg behind the headline (transports, other decomposition)")
    ap.add_argument("--extras-budget", type=float, default=600.0, help="N > 1: seconds the legs behind the headline may take before rank 0 prints the line it has and every rank leaves")
    ap.add_argument("--dry-transports", action="store_true",
                    help="test hook: the N > 1 control flow (legs, barriers, gathers, watchdog) with sleeps instead of kernels -- no GPU is touched")
    ap.add_argument("--same-device", action="store_true",
                    help="diagnostic for one-GPU boxes: every rank of an N > 1 run uses device 0 (RCCL refuses that, so combine with --default-transport ipc); exercises the whole "
                         "N > 1 control flow, the ipc and local_peer transports and both decompositions on one device")
    ap.add_argument("--default-transport", choices=["rccl", "ipc"], default="rccl", help="N > 1: the transport of the headline leg (`value`); the others are reported under `transports`")
    ap.add_argument("--ipc-helper", action="store_true", help="internal: one of the two parked rank processes of the multi_rank_path leg (see start_ipc_helpers)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="test hook: the ranks only report their launch environment (no GPU is touched, the device-count check is skipped)")
    return ap.parse_args(argv)


def visible_gpus() -> int:
    """Number of HIP devices, counted in a short-lived child so that this process never initialises the GPU."""
    code = "import torch; print('JRX_DEVICE_COUNT', torch.cuda.device_count())"
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        sys.stderr.write("bench.py: device count probe timed out after 600 s\n")
        return -1
    for line in reversed(out.stdout.splitlines()):
        if line.startswith("JRX_DEVICE_COUNT "):
            return int(line.split()[1])
    sys.stderr.write(f"bench.py: device count probe failed (rc {out.returncode}); stderr tail:\n{out.stderr[-2000:]}\n")
    return -1


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """Parent of a `python bench.py --gpus N` call: start the N ranks as child processes, relay rank 0's JSON line."""
    n = args.gpus
    if not (args.dry_launch or args.dry_transports):
        have = visible_gpus()
        if have < 0:
            return 2                      # the probe itself failed: its message says why
        if have < (1 if args.same_device else n):
            sys.stderr.write(f"bench.py: {n} GPUs requested, {have} visible\n")
            return 2
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("MASTER_PORT", str(free_port()))
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env0["WORLD_SIZE"] = env0["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(SCRIPT)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # rank 0's stdout is drained by a thread while all children are polled: a rank that dies takes the others down with it
    # (they would otherwise wait for it in a collective for ever); only the exact processes started here are ever killed
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and failed is None:
            failed = time.time()
        if failed is not None and time.time() - failed > 10.0:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    rd.join(timeout=10.0)
    out0 = buf[0] if buf else ""
    rcs = [p.returncode for p in procs]
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    rc = next((c for c in rcs if c != 0), 0)
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def mem_available_gb() -> float:
    """host memory this job may still take: MemAvailable, capped by what is left of the cgroup's memory.max"""
    avail = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / 1e6
    except OSError:
        return 0.0
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = min(avail, (int(lim) - used) / 1e9)
    except (OSError, ValueError):
        pass
    return avail


def cpu_baseline(n_cpu: int, budget_s: float, min_iters: int = 3):
    """The oracle (CPU restatement, 6 unfused kernels, OpenMP) timed on this host's cores."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as orc
    orc.set_num_threads(orc.usable_cpus())      # affinity mask capped by the cgroup CPU quota (16 CPUs on the GPU boxes' 256-thread hosts)
    from __graft_entry__ import load_package
    jr = load_package()
    from justrelax_jl_amd import checks
    import justrelax_jl_amd.grid as g
    g.finalize_global_grid()
    s = jr.miniapps.solvi3d(n_cpu)
    p = checks.oracle_params3d(orc, s)
    et = orc.compute_maxloc(s.arrays["eta"])
    # NUMA placement: every array re-allocated and first touched by the OpenMP threads that work on its z slabs (numpy had placed all pages
    # on the node of the main thread: 31 GB/s on a 128-thread host instead of what its memory system gives)
    for k in list(s.arrays):
        s.arrays[k] = orc.first_touch(s.arrays[k])
    et = orc.first_touch(et)
    orc.stokes3d_iteration(s.arrays, et, p)          # warm
    t0, it = time.perf_counter(), 0
    while True:
        orc.stokes3d_iteration(s.arrays, et, p)
        it += 1
        el = time.perf_counter() - t0
        if (el > budget_s and it >= min_iters) or it >= 2000:
            break
    g.finalize_global_grid()
    return it / el, it, el, orc.num_threads()


# ------------------------------------------------------------------------------------------------ extra legs (one GPU)
def solve_path(jr, h, st, pt, geo, bcs, ρg, K, G, dt, iters, n, pr):
    """jrx_stokes3d_solve itself with the reference's cadence (SolVi3D.jl:119-120: nout = 100): compute_maxloc!, the norm checks with
    their Σx² reductions and host syncs, the un-fused observable iterations and the τ -> τ_o copy are all inside the timed call."""
    import torch
    pt.ϵ_rel = pt.ϵ_abs = 1e-300       # never converge: exactly iterMax + 1 iterations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = jr.solve_(st, pt, geo, bcs, ρg, K, G, dt, None, kwargs=dict(iterMax=iters, nout=100, verbose=False), handle=h)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    cells = float(n) ** 3
    return {"entry": "jrx_stokes3d_solve", "iterMax": iters, "nout": 100, "iterations": int(r.iter), "checks": int(len(r.err_evo1)),
            "it_per_s": r.iter / el, "ms_per_iteration": el / r.iter * 1e3, "device_loop_s": r.time,
            "form": pr["form"], "bytes_per_cell": pr["alg"],
            "effective_GBps": pr["alg"] * cells * r.iter / el / 1e9,
            "frac_of_peak": pr["alg"] * cells * r.iter / el / 1e9 / HBM_PEAK_GBS,
            "norm_Rx_last": float(r.norm_Rx[-1]) if len(r.norm_Rx) else None}


def state_tensors(obj, _seen=None):
    """every device array reachable from a StokesArrays-like object (its fields, nested)"""
    import torch
    seen = _seen if _seen is not None else set()
    if torch.is_tensor(obj):
        if obj.data_ptr() not in seen and obj.numel() > 0:
            seen.add(obj.data_ptr())
            yield obj
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from state_tensors(v, seen)
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from state_tensors(v, seen)
    elif hasattr(obj, "__dict__"):
        for v in vars(obj).values():
            yield from state_tensors(v, seen)
    elif hasattr(obj, "__slots__"):
        for k in obj.__slots__:
            yield from state_tensors(getattr(obj, k, None), seen)


def uniform_chunk_mib(n):
    """one chunk size for every large array of an n^3 block: the largest of them, (n + 2)^2 (n + 1) doubles, rounded up to 2 MiB -- with every array ONE chunk of one common size the
    placement search can deal chunks of a pool that spans the device's memory (csrc/fieldpool.hip, "field_pool_pct")"""
    b = (n + 2) * (n + 2) * (n + 1) * 8
    return -(-b // (2 << 20)) * 2


class PoolArrays:
    """with PoolArrays(h, n): the constructors of the package hand out arrays of the library (jrx_field_alloc) with option "field_placement" = 1 and one chunk size for every large
    array of an n^3 block -- each array ONE physical chunk picked at random from a pool that spans "field_pool_pct" % of the free memory, mapped once at a virtual range never used
    before (csrc/fieldpool.hip).  `on` False: torch's arrays (hipMalloc), nothing changes.  trim(): the pool's unused chunks go back to the driver (after the first driver call has
    made the library's own second state set)."""

    def __init__(self, h, n, on=True, pool_pct=None):
        self.h, self.n, self.on, self.pool_pct = h, n, on, pool_pct

    def __enter__(self):
        if self.on:
            from justrelax_jl_amd import arrays as _arrays
            self.h.set_option("field_placement", 1)
            self.h.set_option("field_chunk_mib", uniform_chunk_mib(self.n))
            if self.pool_pct is not None:
                self.h.set_option("field_pool_pct", self.pool_pct)
            _arrays.use_library_arrays(self.h)
        return self

    def __exit__(self, *exc):
        if self.on:
            from justrelax_jl_amd import arrays as _arrays
            _arrays.use_library_arrays(None)
        return False

    def trim(self):
        if self.on:
            self.h.call("jrx_field_trim")

    def done(self):
        """the arrays of the run have been dropped by the caller: the library's later arrays are hipMalloc's again"""
        if self.on:
            import torch
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            self.h.call("jrx_field_trim")
            self.h.set_option("field_placement", 0)


def state_checksums(st):
    """order-independent, exact: the wrapping int64 sum of the bit patterns of P, the six stresses and the three velocities"""
    import torch
    ts = [st.P, st.τ.xx, st.τ.yy, st.τ.zz, st.τ.yz, st.τ.xz, st.τ.xy, st.V.Vx, st.V.Vy, st.V.Vz]
    return [int(t.contiguous().view(torch.int64).sum().item()) for t in ts]


def cfg_solvi(jr, h, n, steps, warm, pool=True):
    """SolVi3D at n^3 through the same timed batch as the headline (BASELINE configs[2] at n = 256)."""
    import torch
    import justrelax_jl_amd.grid as grid
    from justrelax_jl_amd import stokes
    from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
    grid.finalize_global_grid()
    grid.init_global_grid(n, n, n, rank=0, nprocs=1)
    pa = PoolArrays(h, n, pool)
    try:
        with pa:
            st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
            jr.flow_bcs_(st, bcs, handle=h)
            ητ = jr.fzeros((n, n, n), st.P.device)
            jr.compute_maxloc_(ητ, st.viscosity.η, handle=h)
        run = lambda k: stokes.iterate_timed_(st, pt, geo, bcs, ρg, K, G, ητ, dt, k, handle=h)
        run(warm)
        pa.trim()
        torch.cuda.synchronize()
        f0 = counters(h)
        t0 = time.perf_counter()
        tot_ms, sa, sb, sf, sk, kcells = run(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    finally:
        st = ρg = K = G = ητ = None
        pa.done()
    cells = float(n) ** 3
    pr = pricing(h, dt, nof_ran(h, f0))
    out = {"workload": f"SolVi3D {n}^3", "form": pr["form"], "bytes_per_cell": pr["alg"], "steps": steps, "it_per_s": steps / el, "ms_per_step": el / steps * 1e3,
           "frac_whole_iteration": pr["alg"] * cells * steps / el / 1e9 / HBM_PEAK_GBS, "arrays": "pool" if pool else "hipMalloc"}
    if sk > 0:
        out["kernel"] = "k_fused3d"
        out["avg_launch_ms"] = sk
        out["kernel_cells_per_launch"] = kcells
        out["frac_kernel"] = pr["alg"] * kcells / (sk * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["needed_bytes_per_cell"] = pr["needed"]
        out["frac_kernel_at_needed_bytes"] = pr["needed"] * kcells / (sk * 1e-3) / 1e9 / HBM_PEAK_GBS
    grid.finalize_global_grid()
    return out


def _timed(fn, warm, iters):
    import torch
    fn(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(iters)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


def cfg_solcx(jr, h, n=512, iters=2000):
    """SolCx 2D visco-elastic (BASELINE configs[1]); floor 30 passes = 240 B/cell-iteration."""
    from justrelax_jl_amd.miniapps.common import upload_stokes
    s = jr.miniapps.solcx2d(n, iterMax=iters - 1, nout=10 ** 9)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st, ρg, K, G = upload_stokes(s, jr.AMDGPUBackend)
    run = lambda k: jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, G, K, s.dt, None, kwargs=dict(iterMax=k - 1, nout=10 ** 9, verbose=False), handle=h)
    el, r = _timed(run, 50, iters)
    return {"workload": f"SolCx {n}^2 (2D visco-elastic)", "iterations": int(r.iter), "it_per_s": r.iter / el,
            "effective_GBps_at_240B_per_cell": 240.0 * n * n * r.iter / el / 1e9}


def cfg_shearband(jr, h, n=1024, iters=600):
    """2D multiphase visco-elasto-plastic shear band (BASELINE configs[4]); as written ~88 passes = 700 B/cell-iteration."""
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    s = jr.miniapps.shearband2d(n, iterMax=iters - 1, nout=10 ** 9)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    st.V.Vx.copy_(from_numpy(s.arrays["Vx"], dev)); st.V.Vy.copy_(from_numpy(s.arrays["Vy"], dev))
    st.viscosity.η.copy_(from_numpy(s.arrays["eta"], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    pr.center.copy_(from_numpy(s.arrays["phase_c"], dev)); pr.vertex.copy_(from_numpy(s.arrays["phase_v"], dev))
    ρg = (jr.fzeros(s.ni, dev), jr.fzeros(s.ni, dev))
    run = lambda k: jr.solve_(st, s.pt, s.grid, s.flow_bcs, ρg, pr, s.extra["phases"], None, s.dt, None,
                              kwargs=dict(iterMax=k - 1, nout=10 ** 9, iterMin=10 ** 9, verbose=False), handle=h)
    el, r = _timed(run, 30, iters)
    return {"workload": f"shear band {n}^2 (2D multiphase VEP)", "iterations": int(r.iter), "it_per_s": r.iter / el,
            "effective_GBps_at_700B_per_cell_as_written": 700.0 * n * n * r.iter / el / 1e9,
            "needed_bytes_per_cell": 440.0, "effective_GBps_at_needed_bytes": 440.0 * n * n * r.iter / el / 1e9,          # floor ~ 55 passes
            "frac_at_needed_bytes": 440.0 * n * n * r.iter / el / 1e9 / 8000.0}


def cfg_thermal2d(jr, h, n=256, iters=4000):
    """2D PT heat diffusion, array-coefficient form (BASELINE configs[0]); 18 passes = 144 B/cell-iteration."""
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    from justrelax_jl_amd.miniapps.thermal2d import add_perturbation
    dev = torch.device("cuda", torch.cuda.current_device())
    s = jr.miniapps.diffusion2d(n, iterMax=iters, nout=10 ** 9)
    th = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    add_perturbation(s.arrays["T"], s.grid, **s.extra["perturbation"])
    th.T.copy_(from_numpy(s.arrays["T"], dev)); th.H.copy_(from_numpy(s.arrays["H"], dev))
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-300)

    def run(k):
        jr.heatdiffusion_PT_(th, pt, s.flow_bcs, K, ρCp, s.dt, s.grid, kwargs=dict(iterMax=k, nout=10 ** 9, verbose=False), handle=h)
        return k
    el, k = _timed(run, 100, iters)
    return {"workload": f"thermal diffusion {n}^2 (2D PT, array form)", "iterations": k, "it_per_s": k / el,
            "effective_GBps_at_144B_per_cell_as_written": 144.0 * n * n * k / el / 1e9,
            # the one-launch iteration needs 10 reads (T, q(2), K, θr_dτ, Told, ρCp, dτ_ρ, H, SH) + 3 writes = 104 B/cell; 256^2 is cache-resident: a rate, not a roofline fraction
            "needed_bytes_per_cell": 104.0, "effective_GBps_at_needed_bytes": 104.0 * n * n * k / el / 1e9}


def cfg_shearband3d(jr, h, n=256, iters=60):
    """3D multiphase visco-elasto-plastic shear band (Stokes3D.jl:447-668 as test/test_shearband3D_MPI.jl drives it).  Algorithmic traffic of
    one iteration in the reference's kernel decomposition = 114 array passes = 912 B/cell (DESIGN.md, 3D VEP table)."""
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    s = jr.miniapps.shearband3d(n, iterMax=iters - 1, nout=10 ** 9)
    s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
    st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
    for k, t in dict(Vx=st.V.Vx, Vy=st.V.Vy, Vz=st.V.Vz, eta=st.viscosity.η).items():
        t.copy_(from_numpy(s.arrays[k], dev))
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    for k, name in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
        getattr(pr, name).copy_(from_numpy(s.arrays[k], dev))
    phases, grid_, pt, bcs, dt = s.extra["phases"], s.grid, s.pt, s.flow_bcs, s.dt
    del s
    ρg = tuple(jr.fzeros(st._ni, dev) for _ in range(3))
    run = lambda k: jr.solve_(st, pt, grid_, bcs, ρg, pr, phases, None, dt, None, kwargs=dict(iterMax=k - 1, nout=10 ** 9, verbose=False), handle=h)
    el, r = _timed(run, 5, iters)
    gbps = 912.0 * n ** 3 * r.iter / el / 1e9
    return {"workload": f"shear band {n}^3 (3D multiphase VEP)", "iterations": int(r.iter), "it_per_s": r.iter / el,
            "effective_GBps_at_912B_per_cell_as_written": gbps, "frac_of_8TBps_as_written": gbps / 8000.0,
            # what the three kernels of an unobserved iteration have to move, every array once: fused pre / centre kernel 27 reads + 16 writes, edge pass 29 reads (11 centre, 9 shear,
            # 6 phase ratios, 3 λ) + 6 writes, velocity sweep 11 reads + 3 writes (the zero body forces are not loaded) = 92 passes
            "needed_bytes_per_cell": 736.0, "frac_at_needed_bytes": 736.0 * n ** 3 * r.iter / el / 1e9 / 8000.0}


def cfg_thermal3d(jr, h, n=256, iters=400):
    """3D PT heat diffusion, array-coefficient form (DiffusionPT_solver.jl:34-149); 22 passes = 176 B/cell-iteration as the reference's two kernels move
    them (flux: R T, K, θ, q(3) W q(3), q2(3); update: R q(3), Told, ρCp, dτ_ρ, H, SH, T W T)."""
    import torch
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    s = jr.miniapps.diffusion3d(n, iterMax=iters, nout=10 ** 9)
    th = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    th.T.copy_(from_numpy(s.arrays["T"], dev)); th.H.fill_(1.0e-6)
    K, ρCp = from_numpy(s.arrays["K"], dev), from_numpy(s.arrays["rhoCp"], dev)
    pt = jr.PTThermalCoeffs(jr.AMDGPUBackend, K, ρCp, s.dt, s.extra["di"], s.extra["li"], CFL=s.pt["CFL"], ϵ=1e-300)
    bcs, dt, grid_ = s.flow_bcs, s.dt, s.grid
    del s

    def run(k):
        jr.heatdiffusion_PT_(th, pt, bcs, K, ρCp, dt, grid_, kwargs=dict(iterMax=k, nout=10 ** 9, verbose=False), handle=h)
        return k
    el, k = _timed(run, 20, iters)
    gbps = 176.0 * n ** 3 * k / el / 1e9
    return {"workload": f"thermal diffusion {n}^3 (3D PT, array form)", "iterations": k, "it_per_s": k / el,
            "effective_GBps_at_176B_per_cell_as_written": gbps, "frac_of_8TBps_as_written": gbps / 8000.0,
            # the one-launch iteration needs 11 reads (T, q(3), K, θr_dτ, Told, ρCp, dτ_ρ, H, SH) + 4 writes (T, q(3)) = 120 B/cell
            "needed_bytes_per_cell": 120.0, "frac_at_needed_bytes": 120.0 * n ** 3 * k / el / 1e9 / 8000.0}


def cfg_thermal3d_phases(jr, h, n=256, iters=200):
    """3D PT heat diffusion, phase-ratio form (heatdiffusion_PT!(...; kwargs = (phase = phase_ratios, ...)), DiffusionPT_solver.jl:181-305, two phases): per
    iteration update_pt_thermal_arrays! + compute_flux! + update_T!; 35 passes = 280 B/cell-iteration as those three kernels move them (coefficients: R T, P,
    ratios(2) W θ, dτ_ρ; flux: R T, θ, q(3), face ratios 3 x 2 W q(3), q2(3); update: R q(3), Told, T, P, ratios(2), dτ_ρ, H, SH W T)."""
    import torch
    from types import SimpleNamespace
    from justrelax_jl_amd.arrays import from_numpy
    dev = torch.device("cuda", torch.cuda.current_device())
    s = jr.miniapps.diffusion3d_multiphase(n, iterMax=iters, nout=10 ** 9)
    th = jr.ThermalArrays(jr.AMDGPUBackend, s.ni)
    th.T.copy_(from_numpy(s.arrays["T"], dev)); th.H.fill_(1.0e-6)
    pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
    for k, v in s.extra["phase_ratios"].items():
        getattr(pr, k).copy_(from_numpy(v, dev))
    args = SimpleNamespace(P=jr.fzeros(s.ni, dev), T=th.T)
    pt = jr.PTThermalCoeffs.from_phases(jr.AMDGPUBackend, s.extra["rheology"], pr, args, s.dt, s.ni, s.extra["di"], s.extra["li"], ϵ=1e-300, CFL=s.pt["CFL"])
    bcs, dt, grid_, rheo = s.flow_bcs, s.dt, s.grid, s.extra["rheology"]
    del s

    def run(k):
        jr.heatdiffusion_PT_(th, pt, bcs, rheo, args, dt, grid_, kwargs=dict(phase=pr, iterMax=k, nout=10 ** 9, verbose=False), handle=h)
        return k
    el, k = _timed(run, 10, iters)
    gbps = 280.0 * n ** 3 * k / el / 1e9
    return {"workload": f"thermal diffusion {n}^3 (3D PT, phase-ratio form, 2 phases)", "iterations": k, "it_per_s": k / el,
            "effective_GBps_at_280B_per_cell_as_written": gbps, "frac_of_8TBps_as_written": gbps / 8000.0,
            # k_thermal3d_fused_ph, two phases: 18 reads (T, θr_dτ, q(3), P, dτ_ρ, Told, H, SH, centre and three face ratio arrays of 2 doubles each) + 6 writes (T, q(3), θr_dτ, dτ_ρ) = 192 B/cell
            "needed_bytes_per_cell": 192.0, "frac_at_needed_bytes": 192.0 * n ** 3 * k / el / 1e9 / 8000.0}


def cfg_multi_rank_path(jr, n=512, steps=40, warm=6, only=None, splits=("x", "z"), handle_options=None):
    """The N > 1 code path priced on ONE device: two different n^3 blocks of an IGG decomposition (two handles of this process joined by
    jrx_comm_init_local, planes pushed by device-to-device copies, one host thread per rank) run the timed batch of the headline concurrently.
    The same two blocks -- same allocations, the pool's boxes and allocations differ by several per cent -- are then timed again without the
    communicator: `overhead_pct` is what the exchange (BCs in memory, pack, copies, unpack, stress fix-up next to the received planes) costs
    on top.  The two blocks share the device's HBM, so two uncoupled blocks run at the one-block rate (also reported)."""
    import torch
    import justrelax_jl_amd.grid as grid
    from justrelax_jl_amd import _lib, halo, stokes
    from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
    dev = torch.cuda.current_device()
    DIMS = {"x": (2, 1, 1), "y": (1, 2, 1), "z": (1, 1, 2), "xyz": (2, 2, 2)}
    out = {"workload": f"SolVi3D, two {n}^3 blocks on one device (in-process transport: hipMemcpyAsync D2D + events)", "steps": steps}

    ALTERNATIONS = 5

    def stats(pairs):
        """[(coupled, uncoupled) block-it/s, ...] -> medians and spread of the paired overheads"""
        ov = sorted((u / c - 1.0) * 100.0 for c, u in pairs)
        med = lambda v: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2])
        return {"block_it_per_s": med([c for c, _ in pairs]), "uncoupled_block_it_per_s": med([u for _, u in pairs]), "overhead_pct": med(ov), "overhead_pct_min": ov[0],
                "overhead_pct_max": ov[-1], "alternations": len(pairs)}

    def timed(hs, blocks, k=steps):
        fns = lambda m: [(lambda r=r: stokes.iterate_timed_(*blocks[r], m, handle=hs[r])) for r in range(len(hs))]
        halo.run_ranks(fns(warm))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        halo.run_ranks(fns(k))
        torch.cuda.synchronize()
        return len(hs) * k / (time.perf_counter() - t0)

    def split_leg(name, modes, n=n, pool=True):
        """ONE set of handles and arrays: the blocks are timed with the communicator (coupled) and, after jrx_comm_destroy, without it (uncoupled), alternating -- the same caller arrays
        and the same library-owned state sets in both, so that what differs is the exchange and nothing else (two handle sets would also differ in where their own arrays lie, which
        moves the kernel by more than the exchange costs: profiles/r05_placement_search.txt).  The blocks' arrays come from the placement pool of their handles (PoolArrays)."""
        dims = DIMS[name]
        nr = dims[0] * dims[1] * dims[2]
        hs = [_lib.Handle(dev) for _ in range(nr)]
        carts = halo.make_carts((n, n, n), dims)
        blocks, res, coupled = [], {}, [False]

        def couple():
            if not coupled[0]:
                halo.init_comm_local(hs, carts)
                coupled[0] = True
                # the ghost planes of V and ητ before the first iteration (what the drivers do at their start)
                halo.run_ranks([(lambda r=r: halo.update_halo_(blocks[r][0].V.Vx, blocks[r][0].V.Vy, blocks[r][0].V.Vz, blocks[r][7], ni=(n, n, n), handle=hs[r])) for r in range(nr)])

        def uncouple():
            if coupled[0]:
                for h in hs:
                    h.call("jrx_comm_destroy")
                coupled[0] = False

        try:
            for r in range(nr):
                grid.finalize_global_grid()
                grid.init_global_grid(n, n, n, rank=r, nprocs=nr, dimx=dims[0], dimy=dims[1], dimz=dims[2])
                hs[r].set_option("operand_cache", 1)
                for k_, v_ in (handle_options or {}).items():
                    hs[r].set_option(k_, v_)
                with PoolArrays(hs[r], n, pool, pool_pct=70 // nr):
                    st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
                    jr.flow_bcs_(st, bcs, handle=hs[r])
                    ητ = jr.fzeros((n, n, n), st.P.device)
                    jr.compute_maxloc_(ητ, st.viscosity.η, handle=hs[r])
                blocks.append((st, pt, geo, bcs, ρg, K, G, ητ, dt))
                stokes.iterate_timed_(*blocks[r], 2, handle=hs[r])          # the library's second state set comes from the pool as well; then the pool's rest goes back
                hs[r].call("jrx_field_trim")
            # the quoted mode against the same blocks WITHOUT the communicator, alternating: one pair is not evidence (single pairs spanned 0.9 - 14.5 %) -- the
            # median of the paired overheads and their spread are reported
            pairs = []
            for rep in range(ALTERNATIONS if modes[0] == "default" else 1):
                couple()
                for h in hs:
                    h.set_option("fused_overlap", {"serial": 0, "overlap": 1, "early": 2, "default": 3}[modes[0]])
                c = timed(hs, blocks)
                uncouple()
                u = timed(hs, blocks)
                pairs.append((c, u))
            res[modes[0]] = pairs
            for mode in modes[1:]:
                couple()
                for h in hs:
                    h.set_option("fused_overlap", {"serial": 0, "overlap": 1, "early": 2, "default": 3}[mode])
                c = timed(hs, blocks)
                uncouple()
                res[mode] = [(c, timed(hs, blocks))]
            uncouple()
            res["one_block"] = timed(hs[:1], blocks[:1])
            return res
        finally:
            del blocks
            for h in hs:
                h.close()
            torch.cuda.empty_cache()
            grid.finalize_global_grid()

    def vep_leg(name, nv=256, iters=40):
        """the same measurement for jrx_stokes3d_vep_solve (three exchanges per iteration: ητ, the edge stresses, V): two nv^3 shear-band blocks"""
        from justrelax_jl_amd.arrays import from_numpy
        dims = DIMS[name]
        hs = [_lib.Handle(dev) for _ in range(2)]
        hu = [_lib.Handle(dev) for _ in range(2)]           # the same blocks uncoupled
        tdev = torch.device("cuda", dev)
        blocks, res = [], {}
        try:
            halo.init_comm_local(hs, halo.make_carts((nv, nv, nv), dims))
            s = jr.miniapps.shearband3d(nv, iterMax=iters - 1, nout=10 ** 9)
            s.pt.ϵ_rel = s.pt.ϵ_abs = 1e-300
            for r in range(2):
                st = jr.StokesArrays(jr.AMDGPUBackend, s.ni)
                for k, t in dict(Vx=st.V.Vx, Vy=st.V.Vy, Vz=st.V.Vz, eta=st.viscosity.η).items():
                    t.copy_(from_numpy(s.arrays[k], tdev))
                pr = jr.PhaseRatios(jr.AMDGPUBackend, 2, s.ni)
                for k, nm in (("phase_c", "center"), ("phase_yz", "yz"), ("phase_xz", "xz"), ("phase_xy", "xy")):
                    getattr(pr, nm).copy_(from_numpy(s.arrays[k], tdev))
                blocks.append((st, pr, tuple(jr.fzeros(st._ni, tdev) for _ in range(3))))
            phases, grid_, pt, bcs, dt = s.extra["phases"], s.grid, s.pt, s.flow_bcs, s.dt
            del s

            def run(hh, k):
                fns = [(lambda r=r: jr.solve_(blocks[r][0], pt, grid_, bcs, blocks[r][2], blocks[r][1], phases, None, dt, None,
                                              kwargs=dict(iterMax=k - 1, nout=10 ** 9, verbose=False), handle=hh[r])) for r in range(len(hh))]
                halo.run_ranks(fns)

            def timed_vep(hh):
                run(hh, 5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(hh, iters)
                torch.cuda.synchronize()
                return len(hh) * iters / (time.perf_counter() - t0)

            for mode, v in (("serial", 0), ("hidden", 2), ("hidden_eta_tau_only", 1)):
                for h in hs:
                    h.set_option("vep3_hide_comm", v)
                res[mode] = [(timed_vep(hs), timed_vep(hu)) for _ in range(3 if mode == "serial" else 1)]
            res["one_block"] = timed_vep(hu[:1])
            return res
        finally:
            del blocks
            for h in hs + hu:
                h.close()
            torch.cuda.empty_cache()
            grid.finalize_global_grid()

    if only and only[0] == "vep":
        return {"leg": list(only), "block_it_per_s": vep_leg(only[1])}
    if only:       # profiling hook (scripts/bench_multi_rank.py): one coupled leg alone, e.g. ("z", "serial")
        return {"leg": list(only), "block_it_per_s": {k: (stats(v) if isinstance(v, list) and v and isinstance(v[0], tuple) else v) for k, v in split_leg(only[0], [only[1]], n=n).items()}}
    best = None
    for name in splits:
        r = split_leg(name, ["default", "serial", "early", "overlap"])
        leg = {"one_block_it_per_s": r["one_block"], "arrays": "placement pool per block"}
        for mode in ("default", "serial", "early", "overlap"):
            leg[mode] = stats(r[mode])
        leg["two_uncoupled_blocks_block_it_per_s"] = leg["default"]["uncoupled_block_it_per_s"]
        out[f"split_{name}"] = leg
        cand = "default"           # the default pipeline is what the leg quotes (median of the alternations): the kernel's own boundary tiles read the received planes (a second launch of the
                                   # kernel behind the exchange: no BC launch, no fix-up); "early" (exchange beside the kernel, BCs + fix-up behind it), "serial" and "overlap" (shell
                                   # tiles) are the older options, one pair each
        if best is None or leg[cand]["overhead_pct"] > best[1]["overhead_pct"]:
            best = (f"split_{name}/{cand}", leg[cand])          # the headline of the leg is the WORSE split (x planes are strided)
    out["it_per_s"] = best[1]["block_it_per_s"]
    out["overhead_pct"] = best[1]["overhead_pct"]
    out["quoted"] = best[0]
    try:
        r = vep_leg("z")
        out["vep3d_256_split_z"] = {"workload": "jrx_stokes3d_vep_solve, two 256^3 shear-band blocks, exchanges of ητ, the edge stresses and V every iteration",
                                    "one_block_it_per_s": r["one_block"], "serial": stats(r["serial"]), "hidden": stats(r["hidden"]), "hidden_eta_tau_only": stats(r["hidden_eta_tau_only"]),
                                    "default": "serial (tuning switch vep3_hide_comm = 0): the median of three alternations against the same blocks uncoupled; the two hidden forms one pair each"}
        out["vep3d_256_split_z"]["two_uncoupled_blocks_block_it_per_s"] = out["vep3d_256_split_z"]["serial"]["uncoupled_block_it_per_s"]
    except Exception as e:
        out["vep3d_256_split_z"] = {"error": f"{type(e).__name__}: {e}"}
    # the same two blocks as two PROCESSES on this device through the cross-process copy-engine transport (this process is idle meanwhile)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    out["ipc_two_processes"] = run_ipc_helpers()
    # What the leg quotes: one PROCESS per block (the deployment of the reference, test/runtests.jl:73-90 mpiexec -n 2, and of the Julia extension) when that ran; the in-process figures above
    # come from two host threads of one interpreter and carry their scheduling noise (4 - 17 % between runs of the same pipeline)
    try:
        ipc = out["ipc_two_processes"]
        worst = max(((ipc[f"split_{n_}"]["default"]["overhead_pct"], f"ipc_two_processes/split_{n_}/default", ipc[f"split_{n_}"]["default"]) for n_ in splits if f"split_{n_}" in ipc),
                    key=lambda t: t[0])
        out["in_process_quoted"] = {"quoted": out["quoted"], "it_per_s": out["it_per_s"], "overhead_pct": out["overhead_pct"]}
        out["it_per_s"], out["overhead_pct"], out["quoted"] = worst[2]["block_it_per_s"], worst[0], worst[1]
    except Exception:
        pass
    return out


def other_configs(jr, h):
    import justrelax_jl_amd.grid as grid
    out = {}
    for key, fn in (("multi_rank_path", lambda: cfg_multi_rank_path(jr)),
                    ("solvi3d_256", lambda: cfg_solvi(jr, h, 256, 200, 20)), ("solcx_512", lambda: cfg_solcx(jr, h)),
                    ("shearband_1024", lambda: cfg_shearband(jr, h)), ("thermal2d_256", lambda: cfg_thermal2d(jr, h)),
                    ("shearband3d_256", lambda: cfg_shearband3d(jr, h)), ("thermal3d_256", lambda: cfg_thermal3d(jr, h)),
                    ("thermal3d_phases_256", lambda: cfg_thermal3d_phases(jr, h))):
        try:
            grid.finalize_global_grid()
            out[key] = fn()
        except Exception as e:      # a failing side leg must not lose the headline line; it is reported, not hidden
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    grid.finalize_global_grid()
    return out


# ------------------------------------------------------------------------------------------------ N > 1: transports and decompositions
CHAIN_KEYS = ("k_fused3d", "slab_velocity", "flow_bcs_pre", "update_halo", "flow_bcs_post", "fixup", "step", "beyond_kernel")
YZ_DIMS = {2: (1, 1, 2), 4: (1, 2, 2), 8: (1, 2, 4), 16: (1, 4, 4)}


class Control:
    """the control plane of the ranks (gloo over loopback): barriers, max-reduce, object gathers; a no-op for one rank"""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def max(self, vals):
        if self.world == 1:
            return list(vals)
        import torch
        import torch.distributed as dist
        t = torch.tensor(list(vals), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out


def collective_leg(R, ctl, transport, steps, warm, n1_rate=None):
    """One transport with one process per GPU, all ranks: connect, warm up, time `steps` iterations between barriers (max over ranks), then a short profiled
    batch for the per-rank chain breakdown.  A rank that fails reports it to the others (every stage ends in a gather), so the leg ends as an error entry
    instead of a hang."""
    def stage(fn):
        try:
            r, err = fn(), None
        except Exception as e:      # noqa: BLE001 -- reported in the JSON line
            r, err = None, f"{type(e).__name__}: {e}"
        errs = [e for e in ctl.gather(err) if e]
        return r, (errs[0] if errs else None)

    _, err = stage(lambda: R.connect(transport))
    if err:
        return {"error": f"connect: {err}"}
    leg = {"ranks": R.comm_count(), "steps": steps}

    def timed():
        R.run(warm)
        R.sync(); ctl.barrier()
        t0 = time.perf_counter()
        r = R.run(steps)
        R.sync()
        el = time.perf_counter() - t0
        ctl.barrier()
        return el, r
    res, err = stage(timed)
    if err:
        leg["error"] = f"timed batch: {err}"
        return leg
    el = ctl.max([res[0]])[0]
    leg.update(it_per_s=ctl.world * steps / el, ms_per_step=el / steps * 1e3, pipeline=R.pipeline())
    if n1_rate:
        leg["efficiency_vs_n1"] = (steps / el) / n1_rate
    ch, err = stage(lambda: R.chain(12))
    leg["chain_us_per_rank"] = ctl.gather(ch) if not err else {"error": err}
    return leg


class GpuRanks:
    """what the legs need of one rank: the SolVi3D block of this rank on its device, a communicator of a given transport, timed batches"""

    def __init__(self, args, jr, rank, world, local_rank):
        import torch
        from justrelax_jl_amd import _lib
        self.args, self.jr, self.rank, self.world, self.local_rank = args, jr, rank, world, local_rank
        self.n = args.n
        self.dev = torch.device("cuda", local_rank)
        self.h = _lib.default_handle(local_rank)
        self.blk = None
        self.transport = None
        self.dims = None
        self.placement = None

    def grid_dims(self, mode):
        if mode in ("x", "y", "z"):
            return tuple(self.world if c == mode else 1 for c in "xyz")
        return YZ_DIMS.get(self.world, (1, 1, self.world)) if mode == "yz" else None

    def build(self, mode):
        """the global grid of the decomposition `mode` (balanced | yz); the communicator and the block are dropped -- the next connect() builds this rank's block of
        SolVi3D on it (the ten smoothing passes of the viscosity exchange their halos, SolVi3D.jl:33-40, so the communicator comes first)"""
        import justrelax_jl_amd.grid as grid
        n = self.n
        self.disconnect()
        self.blk = None
        grid.finalize_global_grid()
        d = self.grid_dims(mode)
        if d:
            grid.init_global_grid(n, n, n, rank=self.rank, nprocs=self.world, dimx=d[0], dimy=d[1], dimz=d[2])
        else:
            grid.init_global_grid(n, n, n, rank=self.rank, nprocs=self.world)
        self.dims = tuple(grid.global_grid().dims)
        return self.dims

    def disconnect(self):
        if self.transport:
            self.h.call("jrx_comm_destroy")
            self.transport = None

    def connect(self, transport):
        from justrelax_jl_amd import halo
        from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
        self.disconnect()
        self.h.set_option("comm_timeout_ms", 60000)
        if transport == "rccl":
            halo.init_comm(self.h)
        elif transport == "ipc":
            halo.init_comm_ipc(self.h)
        else:
            raise ValueError(transport)
        self.transport = transport
        n = self.n
        fresh = self.blk is None
        if fresh:
            # as at N = 1: the block's arrays come from the placement pool of the rank's handle (PoolArrays) -- on every rank or on none
            pool = self.args.placement == "pool" and self._pool_everywhere()
            uh = lambda a: halo.update_halo_(a, ni=(n, n, n), handle=self.h)
            with PoolArrays(self.h, n, pool, pool_pct=(70 // self.world if self.args.same_device else None)):
                st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, self.jr.AMDGPUBackend, update_halo=uh)
                self.jr.flow_bcs_(st, bcs, handle=self.h)
                ητ = self.jr.fzeros((n, n, n), self.dev)
                self.jr.compute_maxloc_(ητ, st.viscosity.η, handle=self.h)
            self.blk = (st, pt, geo, bcs, ρg, K, G, ητ, dt)
            self.placement = "pool" if pool else "hipMalloc"
        st, ητ = self.blk[0], self.blk[7]
        halo.update_halo_(st.V.Vx, st.V.Vy, st.V.Vz, ητ, ni=(n, n, n), handle=self.h)
        if fresh and self.placement == "pool":
            self.run(2)                                   # the library's second state set comes from the pool as well; then the pool's rest goes back to the driver
            self.sync()
            self.h.call("jrx_field_trim")

    def _pool_everywhere(self):
        """the placement pool needs the driver's virtual-memory-management calls: a rank where they are refused keeps torch's arrays, and then so do all (a weak-scaling run is
        paced by its slowest rank; ranks that differ in where their arrays come from would not be one configuration)"""
        import torch
        import torch.distributed as dist
        from justrelax_jl_amd import arrays as _arrays
        ok = 1.0
        try:
            self.h.set_option("field_placement", 1)
            self.h.set_option("field_chunk_mib", 2)
            _arrays.use_library_arrays(self.h)
            t = self.jr.fzeros((64, 64, 600), self.dev)      # 19.7 MB: chunk-backed (2 MiB chunks: no pool yet)
            del t
        except Exception as e:      # noqa: BLE001
            ok = 0.0
            sys.stderr.write(f"bench: rank {self.rank}: library arrays refused ({type(e).__name__}: {e})\n")
        finally:
            _arrays.use_library_arrays(None)
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            t_ok = torch.tensor([ok])
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            ok = float(t_ok.item())
        if ok == 0.0:
            self.h.set_option("field_placement", 0)
        return ok == 1.0

    def comm_count(self):
        import ctypes as C
        cnt = C.c_int32(0)
        self.h.call("jrx_comm_count", C.byref(cnt))
        return cnt.value

    def pipeline(self):
        return {0: "exchange behind the kernel", 1: "shell tiles + exchange beside the interior tiles", 2: "early exchange beside the kernel, BCs + fix-up behind it",
                3: "exchange beside the kernel, whose boundary tiles read the received planes (second launch behind the exchange, no BCs, no fix-up)",
                4: "exchange beside the kernel, whose boundary tiles read the received planes (second launch behind the exchange, no BCs, no fix-up)"}[self.h.get_option("fused_overlap")]

    def run(self, k):
        from justrelax_jl_amd import stokes
        return stokes.iterate_timed_(*self.blk, k, handle=self.h)

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def chain(self, k):
        return chain_profile(self.h, lambda: self.run(k))


def chain_profile(h, run):
    """per-stage microseconds of a rank's fused iteration (jrx_tuning_chain_profile; hipEvents inside the library on the streams the stages run on)"""
    import ctypes as C
    h.set_option("chain_profile", 1)
    try:
        run()
        out, ns = (C.c_double * 8)(), C.c_int64(0)
        h.check(h.lib.jrx_tuning_chain_profile(h._h, out, C.byref(ns)))
    finally:
        h.set_option("chain_profile", 0)
    d = {k: round(out[i], 1) for i, k in enumerate(CHAIN_KEYS)}
    d["samples"] = ns.value
    return d


def local_peer_leg(jr, args, world, mode, steps, warm, n1_rate=None):
    """ONE process (this one) driving `world` handles on `world` devices: jrx_comm_init_local, planes pushed by hipMemcpyPeerAsync (copy engines over xGMI),
    one host thread per rank.  The other rank processes idle meanwhile."""
    import torch
    import justrelax_jl_amd.grid as grid
    from justrelax_jl_amd import _lib, halo, stokes
    from justrelax_jl_amd.miniapps.stokes3d import solvi3d_device
    n = args.n
    d = YZ_DIMS.get(world, (1, 1, world)) if mode == "yz" else None
    cur = torch.cuda.current_device()
    devs = [0] * world if args.same_device else list(range(world))
    hs, blocks = [], []
    try:
        for r in range(world):
            torch.cuda.set_device(devs[r])
            hs.append(_lib.Handle(devs[r]))
        torch.cuda.set_device(cur)
        grid.finalize_global_grid()
        if d:
            grid.init_global_grid(n, n, n, rank=0, nprocs=world, dimx=d[0], dimy=d[1], dimz=d[2])
        else:
            grid.init_global_grid(n, n, n, rank=0, nprocs=world)
        dims = tuple(grid.global_grid().dims)
        halo.init_comm_local(hs, halo.make_carts((n, n, n), dims))
        for r in range(world):
            torch.cuda.set_device(devs[r])
            grid.finalize_global_grid()
            grid.init_global_grid(n, n, n, rank=r, nprocs=world, dimx=dims[0], dimy=dims[1], dimz=dims[2])
            st, ρg, K, G, pt, geo, bcs, dt = solvi3d_device(n, jr.AMDGPUBackend)
            jr.flow_bcs_(st, bcs, handle=hs[r])
            ητ = jr.fzeros((n, n, n), st.P.device)
            jr.compute_maxloc_(ητ, st.viscosity.η, handle=hs[r])
            blocks.append((st, pt, geo, bcs, ρg, K, G, ητ, dt))

        def on(r, fn):
            def f():
                torch.cuda.set_device(devs[r])    # the current device is a property of the host thread
                return fn()
            return f
        halo.run_ranks([on(r, lambda r=r: halo.update_halo_(blocks[r][0].V.Vx, blocks[r][0].V.Vy, blocks[r][0].V.Vz, blocks[r][7], ni=(n, n, n), handle=hs[r])) for r in range(world)])

        def sync_all():
            for dv in set(devs):
                torch.cuda.synchronize(dv)
        batch = lambda k: halo.run_ranks([on(r, lambda r=r: stokes.iterate_timed_(*blocks[r], k, handle=hs[r])) for r in range(world)])
        batch(warm)
        sync_all()
        t0 = time.perf_counter()
        batch(steps)
        sync_all()
        el = time.perf_counter() - t0
        leg = {"handles": world, "devices": devs, "steps": steps, "decomposition": list(dims), "it_per_s": world * steps / el, "ms_per_step": el / steps * 1e3,
               "pushed_by": "hipMemcpyPeerAsync between the handles of one process, ordered by events"}
        if n1_rate:
            leg["efficiency_vs_n1"] = (steps / el) / n1_rate
        for h in hs:
            h.set_option("chain_profile", 1)
        batch(12)
        sync_all()
        import ctypes as C
        per = []
        for h in hs:
            out, ns = (C.c_double * 8)(), C.c_int64(0)
            h.check(h.lib.jrx_tuning_chain_profile(h._h, out, C.byref(ns)))
            per.append(dict({k: round(out[i], 1) for i, k in enumerate(CHAIN_KEYS)}, samples=ns.value))
        leg["chain_us_per_rank"] = per
        return leg
    finally:
        del blocks
        for h in hs:
            h.close()
        torch.cuda.set_device(cur)
        torch.cuda.empty_cache()
        grid.finalize_global_grid()


class Watchdog:
    """The legs behind the headline talk to other processes; if one of them hangs, rank 0 still prints the line it has (with the leg marked) and every rank
    leaves -- the measured headline is never lost to an extra."""

    def __init__(self, seconds, rank, emit):
        import threading
        self.where, self.rank, self.emit = "start", rank, emit
        self.t = threading.Timer(seconds, self.fire)
        self.t.daemon = True
        self.t.start()

    def fire(self):
        sys.stderr.write(f"bench.py: rank {self.rank}: the extra legs exceeded their time budget during `{self.where}`; leaving\n")
        if self.rank == 0:
            self.emit(f"time budget exceeded during `{self.where}`")
        # The headline (measured, and printed just above with `extras_incomplete` and `degraded` set) is complete: the exit code stays 0 so that a launcher which discards the
        # output of a failed job keeps it -- a reader tells a degraded run by those two keys.  (Ranks that leave here skip jrx_comm_destroy; their peers see the group fail after
        # the transport's time-out.)  No re-exec, no child processes from here.
        os._exit(0)

    def cancel(self):
        self.t.cancel()


def multi_rank_extras(R, ctl, args, out, local_peer, wd):
    """transports x decompositions behind the headline leg (which ran on `rccl` with the balanced decomposition)"""
    steps, warm = args.leg_steps, max(args.warmup, 4)
    tr = out["transports"]
    # one block alone on rank 0 (no communicator, same allocations): the N = 1 rate the efficiencies refer to
    wd.where = "n1_reference"
    R.disconnect()
    n1 = None
    if ctl.rank == 0:
        R.run(warm); R.sync()
        t0 = time.perf_counter()
        R.run(steps); R.sync()
        n1 = steps / (time.perf_counter() - t0)
    ctl.barrier()
    n1 = ctl.max([n1 or 0.0])[0]
    out["n1_reference"] = {"it_per_s": n1, "what": "rank 0's block alone, no communicator, same allocations, the other ranks idle"}
    T0 = out["default_transport"]
    T1 = "ipc" if T0 == "rccl" else "rccl"
    if "it_per_s" in tr[T0]:
        tr[T0]["efficiency_vs_n1"] = (tr[T0]["it_per_s"] / ctl.world) / n1
    wd.where = f"{T0} chain profile"
    try:
        R.connect(T0)
        ch = R.chain(12)
    except Exception as e:      # noqa: BLE001
        ch = {"error": f"{type(e).__name__}: {e}"}
    tr[T0]["chain_us_per_rank"] = ctl.gather(ch)
    wd.where = T1
    tr[T1] = collective_leg(R, ctl, T1, steps, warm, n1)
    R.disconnect()
    ctl.barrier()
    wd.where = "local_peer"
    if ctl.rank == 0:
        try:
            tr["local_peer"] = local_peer(ctl.world, "balanced", steps, warm, n1)
        except Exception as e:      # noqa: BLE001
            tr["local_peer"] = {"error": f"{type(e).__name__}: {e}"}
    ctl.barrier()
    # the other decomposition (x, the contiguous direction, never split) on the faster of the process-per-GPU transports
    wd.where = "alt_decomposition"
    rate = lambda k: tr.get(k, {}).get("it_per_s", 0.0)
    best = "ipc" if rate("ipc") > rate("rccl") else "rccl"
    alt_mode = "balanced" if args.dims == "yz" else "yz"
    try:
        dims = R.build(alt_mode)
        err = None
    except Exception as e:      # noqa: BLE001
        dims, err = None, f"{type(e).__name__}: {e}"
    errs = [e for e in ctl.gather(err) if e]
    if errs:
        out["alt_decomposition"] = {"error": errs[0]}
    else:
        leg = collective_leg(R, ctl, best, steps, warm, n1)
        leg.update(decomposition=list(dims), transport=best)
        out["alt_decomposition"] = leg
        R.disconnect()
        ctl.barrier()
        wd.where = "alt_decomposition local_peer"
        if ctl.rank == 0:
            try:
                leg["local_peer"] = local_peer(ctl.world, alt_mode, steps, warm, n1)
            except Exception as e:      # noqa: BLE001
                leg["local_peer"] = {"error": f"{type(e).__name__}: {e}"}
        ctl.barrier()


# ------------------------------------------------------------------------------------------------ dry run of the N > 1 control flow (no GPU)
class DryRanks:
    """stands in for GpuRanks in `--dry-transports`: the legs' control flow (connects, barriers, gathers, watchdog) runs with sleeps instead of kernels, so
    that the schema of the N > 1 line can be tested without a GPU (tests/test_bench_launch.py)"""

    def __init__(self, args, rank, world):
        self.rank, self.world, self.transport, self.dims, self.n = rank, world, None, None, args.n
        self.fail = os.environ.get("JRX_DRY_FAIL", "")

    def build(self, mode):
        self.disconnect()
        self.dims = YZ_DIMS.get(self.world, (1, 1, self.world)) if mode == "yz" else {2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(self.world, (self.world, 1, 1))
        return self.dims

    def disconnect(self):
        self.transport = None

    def connect(self, transport):
        if transport == self.fail and self.rank == self.world - 1:
            raise RuntimeError(f"dry run: {transport} refused")
        if self.fail == "hang:" + transport and self.rank == self.world - 1:
            time.sleep(3600.0)
        self.transport = transport

    def comm_count(self):
        return self.world

    def pipeline(self):
        return "dry"

    def run(self, k):
        time.sleep(0.001 * k)
        return (1.0 * k, 0.0, 0.0, 1.0, 0.9, float(self.n) ** 3)

    def sync(self):
        pass

    def chain(self, k):
        return dict({key: 1.0 for key in CHAIN_KEYS}, samples=k)


def dry_local_peer(world, mode, steps, warm, n1):
    return {"handles": world, "devices": list(range(world)), "steps": steps, "it_per_s": 1.0, "ms_per_step": 1.0, "efficiency_vs_n1": 1.0,
            "chain_us_per_rank": [dict({k: 1.0 for k in CHAIN_KEYS}, samples=12) for _ in range(world)]}


# ------------------------------------------------------------------------------------------------ N = 1: two rank PROCESSES on the one device (ipc transport)
IPC_HELPERS = []


def start_ipc_helpers(args):
    """Two child processes of this script (`--ipc-helper`), started before this process touches the GPU (afterwards it may not start any) and parked on their stdin until
    the multi_rank_path leg wakes them: rank 0 and rank 1 of a two-block decomposition, both on device 0, joined by jrx_comm_init_ipc."""
    env0 = dict(os.environ, WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", LOCAL_RANK="0")
    for r in range(2):
        opts = [x for kv in args.option for x in ("--option", kv)]
        IPC_HELPERS.append(subprocess.Popen([sys.executable, str(SCRIPT), "--ipc-helper", "--n", str(args.n), "--gpus", "2", *opts], env=dict(env0, RANK=str(r)),
                                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))


def stop_ipc_helpers():
    for p in IPC_HELPERS:
        try:
            if p.stdin and not p.stdin.closed:
                p.stdin.close()
        except OSError:
            pass
    t0 = time.time()
    for p in IPC_HELPERS:
        while p.poll() is None and time.time() - t0 < 20.0:
            time.sleep(0.1)
        if p.poll() is None:
            p.kill()             # only the exact processes started above
    IPC_HELPERS.clear()


def run_ipc_helpers(timeout=420.0):
    """wake the two parked rank processes, wait for rank 0's JSON line; this process must leave the GPU idle meanwhile"""
    if not IPC_HELPERS:
        return {"skipped": "the rank processes are only started by a one-GPU `python bench.py` run with the extra legs on"}
    import threading
    outs = [None, None]

    def drain(i):
        outs[i] = IPC_HELPERS[i].stdout.read()
    ths = [threading.Thread(target=drain, args=(i,), daemon=True) for i in range(2)]
    for t in ths:
        t.start()
    for p in IPC_HELPERS:
        p.stdin.write("go\n")
        p.stdin.flush()
    t0 = time.time()
    while any(p.poll() is None for p in IPC_HELPERS) and time.time() - t0 < timeout:
        time.sleep(0.2)
    alive = [p.poll() is None for p in IPC_HELPERS]
    stop_ipc_helpers()
    for t in ths:
        t.join(timeout=5.0)
    if any(alive):
        return {"error": f"the rank processes did not finish within {timeout:.0f} s"}
    lines = [l for l in (outs[0] or "").splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else {"error": "rank 0 printed no JSON line"}


def ipc_helper(args) -> int:
    """One of the two parked rank processes (see start_ipc_helpers).  Measures, for a split along x and along z: two coupled n^3 blocks (early exchange = default, and the
    exchange behind the kernel) against the same two blocks uncoupled -- what cfg_multi_rank_path measures for the in-process transport."""
    line = sys.stdin.readline()
    if not line.startswith("go"):
        return 0
    rank = int(os.environ["RANK"])
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from datetime import timedelta
    import torch
    import torch.distributed as dist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group(backend="gloo", rank=rank, world_size=2, timeout=timedelta(minutes=10))
    ctl = Control(rank, 2)
    from __graft_entry__ import load_package
    jr = load_package()
    torch.cuda.set_device(0)
    # two processes share this device: a smaller pool each
    args.same_device = True
    R = GpuRanks(args, jr, rank, 2, 0)
    for kv in args.option:
        k, v = kv.split("=")
        R.h.set_option(k, int(v))
    steps, warm = 40, 6
    out = {"workload": f"SolVi3D, two {args.n}^3 blocks on one device, one PROCESS per block (ipc transport: hipIpcOpenMemHandle + hipMemcpyAsync, flags in shared memory)", "steps": steps}
    try:
        for split in ("x", "z"):
            R.build(split)
            leg = {}

            def uncoupled():
                R.disconnect()
                R.run(warm); R.sync(); ctl.barrier()
                t0 = time.perf_counter()
                R.run(steps); R.sync()
                el = ctl.max([time.perf_counter() - t0])[0]
                ctl.barrier()
                return 2 * steps / el

            med = lambda v: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2])
            # the default pipeline against the same two blocks uncoupled, five alternations: the median of the paired overheads is what the leg quotes
            for mode, ov, reps in (("default", 3, 5), ("early", 2, 1), ("serial", 0, 1)):
                R.h.set_option("fused_overlap", ov)
                rates, uncs, last = [], [], {}
                for _ in range(reps):
                    last = collective_leg(R, ctl, "ipc", steps, warm)
                    if "error" in last or not last.get("it_per_s"):
                        break
                    rates.append(last["it_per_s"])
                    uncs.append(uncoupled())
                leg[mode] = {"block_it_per_s": med(rates) if rates else None, "chain_us_per_rank": last.get("chain_us_per_rank"), **({"error": last["error"]} if "error" in last else {})}
                if rates:
                    ov_pct = sorted((u / c - 1.0) * 100.0 for c, u in zip(rates, uncs))
                    leg[mode].update(uncoupled_block_it_per_s=med(uncs), overhead_pct=med(ov_pct), overhead_pct_min=ov_pct[0], overhead_pct_max=ov_pct[-1], alternations=len(rates))
            R.h.set_option("fused_overlap", 3)
            leg["two_uncoupled_blocks_block_it_per_s"] = leg["default"].get("uncoupled_block_it_per_s")
            out[f"split_{split}"] = leg
    except Exception as e:      # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    try:
        R.disconnect()
        dist.barrier()
        dist.destroy_process_group()
    except Exception:           # noqa: BLE001
        pass
    return 0


def check_priced_kernel(h, pr, before, out):
    """Evidence hygiene: the kernel form the line prices must be the one that ran.  The library counts its launches of k_fused3d and, of those, the
    launches of the viscous-limit form; the bench refuses to print a roofline for a form that did not run."""
    now = counters(h)
    d_all, d_visc, d_n1, d_n2 = (b - a for a, b in zip(before[:4], now))
    out["kernel_launch_counters"] = {"k_fused3d": int(d_all), "of_which_viscous_limit_form": int(d_visc), "of_which_without_loads_of_rho_g_x_y": int(d_n1),
                                     "of_which_without_loads_of_any_rho_g": int(d_n2), "operand_checks_failed": int(h.get_option("stat_visc_fallbacks"))}
    want_visc = pr["form"].startswith("viscous_limit")
    want = {0: (0, 0), 1: (d_all, 0), 2: (0, d_all)}[pr.get("nof", 0)]
    if (d_n1, d_n2) != want:
        raise SystemExit(f"bench.py: the line prices the `{pr['form']}` form of k_fused3d, but the launch counters say {out['kernel_launch_counters']}")
    if d_all <= 0 or (want_visc and d_visc != d_all) or (not want_visc and d_visc != 0):
        raise SystemExit(f"bench.py: the line prices the `{pr['form']}` form of k_fused3d, but the library launched {d_all} fused kernels of which {d_visc} in the viscous-limit form")
