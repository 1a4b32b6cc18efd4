"""`python bench.py --gpus N` starts its N ranks itself: child processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set before anything touches a GPU, rank 0's JSON line relayed, non-zero exit when fewer than N devices are visible.
Reference equivalent: `mpiexec -n 2` in test/runtests.jl:73-90."""
import json
import os

import pytest
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_dry_launch_starts_two_ranks(tmp_path):
    env = dict(os.environ, JRX_DRY_LAUNCH_DIR=str(tmp_path))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly one JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["world"] == 2 and out["rank"] == 0
    seen = {}
    for k in (0, 1):
        e = json.loads((tmp_path / f"rank{k}.json").read_text())
        assert e["RANK"] == str(k) and e["LOCAL_RANK"] == str(k) and e["WORLD_SIZE"] == "2"
        assert e["MASTER_ADDR"] == "127.0.0.1" and int(e["MASTER_PORT"]) > 0
        seen[k] = e["MASTER_PORT"]
    assert seen[0] == seen[1]


def test_too_few_devices_fails_cleanly():
    import torch
    have = torch.cuda.device_count()
    want = have + 1 if have >= 1 else 2
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(want)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert f"{want} GPUs requested, {have} visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rank_refuses_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    # with WORLD_SIZE set the script is one rank of an externally launched job (torch.distributed.run): it does not spawn
    assert r.returncode == 0 and json.loads(r.stdout.splitlines()[-1])["world"] == 3


def _strict(line):
    """the contract line: strict JSON (no NaN / Infinity tokens), under 4 KB"""
    assert len(line) < 4096, len(line)
    def no_const(c):
        raise AssertionError(f"non-standard JSON token {c}")
    return json.loads(line, parse_constant=no_const)


def _dry_transports(extra_args=(), env_extra=None, torchrun=False, timeout=180, extras=True, details=None, world=2):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    bench_args = ["--gpus", str(world), "--dry-transports", "--steps", "4", "--warmup", "1", "--leg-steps", "4", *(["--extras"] if extras else []), *(["--details", str(details)] if details else []), *extra_args]
    if torchrun:        # how the driver starts N > 1
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(ROOT / "bench.py"), *bench_args]
    else:
        cmd = [sys.executable, str(ROOT / "bench.py"), *bench_args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def _check_leg(leg, world=2):
    import bench_extras as bench
    assert leg["it_per_s"] > 0 and leg["ms_per_step"] > 0 and 0 < leg["efficiency_vs_n1"]
    per = leg["chain_us_per_rank"]
    assert len(per) == world
    for d in per:
        assert set(bench.CHAIN_KEYS) <= set(d) and d["samples"] > 0


def test_default_two_rank_line_is_compact_and_strict(tmp_path):
    """what the driver reads: ONE line, strict JSON, < 4 KB, with the contract's keys, the per-rank launch times and the name of the details file; without
    --extras no leg behind the headline runs"""
    r, lines = _dry_transports(extras=False, details=tmp_path / "d.json")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout
    out = _strict(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "details"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak" and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert out["config"]["decomposition"] == [2, 1, 1] and "workload" in out["config"]
    assert len(out["roofline"]["launch_ms_per_rank"]) == 2 and out["roofline"]["bound"] == "hbm"
    assert out["details"] == "d.json"
    full = json.loads((tmp_path / "d.json").read_text())
    assert set(full["transports"]) == {"rccl"} and "alt_decomposition" not in full


def test_eight_rank_line_reports_the_2x2x2_rccl_job(tmp_path):
    """BASELINE configs[3] as the driver starts it (torch.distributed.run, 8 ranks): the line reports 8 RCCL ranks, the (2,2,2) decomposition, the 1022^3 global grid
    (2 (512 - 2) + 2 per split dimension) and one launch time per rank -- control flow only, kernels are sleeps"""
    r, lines = _dry_transports(extras=False, details=tmp_path / "d.json", world=8, torchrun=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout
    out = _strict(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["config"]["decomposition"] == [2, 2, 2] and out["config"]["global_grid"] == [1022, 1022, 1022]
    assert len(out["roofline"]["launch_ms_per_rank"]) == 8 and out["scaling"] == "weak"


@pytest.mark.parametrize("torchrun", [False, True])
def test_two_rank_line_carries_every_transport_and_both_decompositions(torchrun, tmp_path):
    """`bench.py --gpus N` (self-launched or under torch.distributed.run, as the driver starts it) prints ONE line whose `value` is the default
    transport (RCCL ranks) and which also carries `transports.{rccl, ipc, local_peer}` -- each with it_per_s, efficiency_vs_n1 and a per-rank chain breakdown -- and an
    `alt_decomposition` leg on the faster process-per-GPU transport.  The control flow (connects, barriers, gathers between two gloo ranks) is the real one; the
    kernels are sleeps (`--dry-transports`), so this runs without a GPU.  Reference: mpiexec -n 2, test/runtests.jl:73-90; docs/paper/paper.md:78-80."""
    sys.path.insert(0, str(ROOT))
    r, lines = _dry_transports(torchrun=torchrun, details=tmp_path / "d.json")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout
    line = _strict(lines[0])
    out = json.loads((tmp_path / "d.json").read_text())            # --extras: the legs behind the headline are in the details file, the line stays the contract
    assert line["value"] == pytest.approx(out["value"], rel=1e-5) and "transports" not in line
    assert out["n_gpus"] == 2 and out["default_transport"] == "rccl" and out["scaling"] == "weak" and out["steps"] == 4
    tr = out["transports"]
    assert set(tr) == {"rccl", "ipc", "local_peer"}
    assert tr["rccl"]["ranks"] == 2 and tr["ipc"]["ranks"] == 2 and tr["local_peer"]["handles"] == 2
    assert tr["rccl"]["it_per_s"] == pytest.approx(out["value"], rel=1e-5)
    for k in tr:
        _check_leg(tr[k])
    alt = out["alt_decomposition"]
    assert alt["decomposition"] == [1, 1, 2] and out["config"]["decomposition"] == [2, 1, 1] and alt["transport"] in ("rccl", "ipc")
    _check_leg(alt)
    _check_leg(alt["local_peer"])
    assert out["n1_reference"]["it_per_s"] > 0 and "extras_incomplete" not in out


def test_a_transport_that_fails_on_one_rank_becomes_an_error_entry(tmp_path):
    r, lines = _dry_transports(env_extra={"JRX_DRY_FAIL": "ipc"}, details=tmp_path / "d.json")
    assert r.returncode == 0, r.stderr[-3000:]
    _strict(lines[-1])
    out = json.loads((tmp_path / "d.json").read_text())
    assert len(lines) == 1 and "refused" in out["transports"]["ipc"]["error"]
    assert out["transports"]["rccl"]["it_per_s"] > 0 and out["transports"]["local_peer"]["it_per_s"] > 0 and out["alt_decomposition"]["transport"] == "rccl"


def test_a_leg_that_hangs_does_not_lose_the_headline():
    """the legs behind the headline run under a time budget: rank 0 prints the line it has, marked, and every rank leaves"""
    r, lines = _dry_transports(extra_args=["--extras-budget", "6"], env_extra={"JRX_DRY_FAIL": "hang:ipc"}, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _strict(lines[-1])
    assert len(lines) == 1 and out["value"] > 0 and "ipc" in out["extras_incomplete"] and out["degraded"] is True


class _FakeHandle:
    def __init__(self, **opts):
        self.opts = opts

    def get_option(self, key):
        return self.opts[key]


def test_bench_prices_a_launch_by_the_kernel_form_that_runs():
    """the viscous-limit form (dt = Inf, option viscous_limit) moves ten operand arrays less: 35 passes priced, 25 needed; any finite dt and the
    option switched off price SURVEY 8d's 45 passes; the roofline object carries the ratios the round-2 verdict asked for (item 6 ii)"""
    sys.path.insert(0, str(ROOT))
    import bench
    h = _FakeHandle(viscous_limit=1, fused_ylds=1)
    v = bench.pricing(h, float("inf"))
    g = bench.pricing(h, 0.25)
    off = bench.pricing(_FakeHandle(viscous_limit=0, fused_ylds=1), float("inf"))
    assert (v["form"], v["alg"], v["needed"]) == ("viscous_limit", 280.0, 200.0)
    assert (g["form"], g["alg"], g["needed"]) == ("general", 360.0, 280.0) and off["form"] == "general"
    n, ms = 512, 6.0
    r = bench.fused_roofline(v, n, ms, ms + 0.1, 0.0, None)
    cells = float(n) ** 3
    assert abs(r["achieved"] - 280.0 * cells / 6.0e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12 and r["frac"] < 1.0
    assert r["bytes_per_cell"] == 280.0 and abs(r["needed_bytes_per_launch"] - 200.0 * cells) < 1.0
    # the PMC figures of the headline form (no body-force loads, NOF = 2: what SolVi3D runs): profiles/pmc_traffic.json, valid while csrc/stokes3d_kernels.hpp is the file they were taken on
    v2 = bench.pricing(h, float("inf"), nof=2)
    assert (v2["form"], v2["alg"], v2["needed"]) == ("viscous_limit_no_body_forces", 256.0, 176.0)
    r2 = bench.fused_roofline(v2, n, ms, ms + 0.1, 0.0, None)
    if bench.X.PMC["stale"]:
        # the kernel source has changed since the PMC passes were taken: the line must not quote them
        assert r2["traffic"] is None and r2["traffic_ratio"] is None and r2["traffic_over_needed"] is None and "STALE" in r2["traffic_source"]
    else:
        assert abs(r2["traffic_ratio"] - bench.X.PMC["k_fused3d_visc_nof2"] / (256.0 * cells)) < 1e-12 and 1.0 < r2["traffic_over_needed"] < 1.6
        assert bench.X.PMC["git_head"] in r2["traffic_source"]
    r256 = bench.fused_roofline(g, 256, 1.0, 1.1, 0.0, None)
    assert r256["traffic"] is None and r256["traffic_ratio"] is None and r256["bytes_per_cell"] == 360.0


def test_pmc_figures_are_dropped_when_the_kernel_source_has_changed(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json carries the sha256 of csrc/stokes3d_kernels.hpp it was measured on; another sha -> every figure None and a source that says STALE"""
    sys.path.insert(0, str(ROOT))
    import bench_extras as bench
    real = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    assert len(real["kernels_sha256"]) == 64 and real["n"] == 512 and real["git_head"]
    fake_root = tmp_path
    (fake_root / "profiles").mkdir()
    (fake_root / "justrelax.jl_amd" / "csrc").mkdir(parents=True)
    (fake_root / "justrelax.jl_amd" / "csrc" / "stokes3d_kernels.hpp").write_text("// another kernel source\n")
    (fake_root / "profiles" / "pmc_traffic.json").write_text(json.dumps(real))
    monkeypatch.setattr(bench, "ROOT", fake_root)
    d = bench.load_pmc()
    assert d["stale"] and d["k_fused3d_visc"] is None and d["k_fused3d_general"] is None and "STALE" in d["source"]
    import hashlib
    real2 = dict(real, kernels_sha256=hashlib.sha256(b"// another kernel source\n").hexdigest())
    (fake_root / "profiles" / "pmc_traffic.json").write_text(json.dumps(real2))
    d = bench.load_pmc()
    assert not d["stale"] and d["k_fused3d_visc"] == real["k_fused3d_visc"] and real["git_head"] in d["source"]


def test_the_contract_line_is_a_fixed_selection_under_4_kb():
    """compact_line: whatever the full record holds (long strings, NaNs, nested legs), the line is strict JSON under 4 KB with the contract's keys"""
    sys.path.insert(0, str(ROOT))
    import bench
    full = {"metric": "m", "value": 200.123456789, "unit": "it/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 5.0, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "w" * 200, "kernel_form": "viscous_limit_no_body_forces", "local_grid": [512] * 3,
                                                                                  "global_grid": [512] * 3, "decomposition": [1, 1, 1], "halo": "none", "arrays": "pool", "operand_cache": 0},
            "state_ok": True, "steady_state": {"steps": 100, "value": 209.0, "ms_per_step": 4.78, "kernel_avg_launch_ms": 4.7, "device_state": {"x": "y" * 5000}},
            "roofline": {"bound": "hbm", "kernel": "k" * 3000, "kernel_name": "k_fused3d", "form": "f", "bytes_per_cell": 256.0, "avg_launch_ms": float("nan"), "achieved": 7200.0, "peak": 8000.0,
                         "unit": "GB/s", "frac": 0.9, "traffic": None, "needed_bytes_per_launch": 2.3e10, "frac_at_needed_bytes": 0.62, "launch_ms_per_rank": [4.7],
                         "general_form": {"bytes_per_cell": 360.0, "avg_launch_ms": 7.3, "frac": 0.82, "what": "z" * 2000}},
            "cpu_baseline": {"value": 2.5, "unit": "it/s", "cores": 16, "kind": "port", "measured_at_n": 256, "sample": "s" * 300, "measured": [{"n": 256}] * 50},
            "other_configs": {"a": "b" * 20000}, "details": "bench_details.json"}
    line = bench.compact_line(full)
    out = _strict(line)
    assert out["roofline"]["avg_launch_ms"] is None and out["roofline"]["general_form"]["frac"] == 0.82 and "other_configs" not in out and "kernel" not in out["roofline"]
    assert out["cpu_baseline"]["cores"] == 16 and out["state_ok"] is True and out["value"] == 200.123
