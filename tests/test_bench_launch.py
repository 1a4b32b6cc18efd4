"""`python bench.py --gpus N` starts its N ranks itself: child processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set before anything touches a GPU, rank 0's JSON line relayed, non-zero exit when fewer than N devices are visible.
Reference equivalent: `mpiexec -n 2` in test/runtests.jl:73-90."""
import json
import os
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
    assert abs(r["traffic_ratio"] - bench.PMC_TRAFFIC_VISC_512 / (280.0 * cells)) < 1e-12 and 1.0 < r["traffic_over_needed"] < 1.6
    r256 = bench.fused_roofline(g, 256, 1.0, 1.1, 0.0, None)
    assert r256["traffic"] is None and r256["traffic_ratio"] is None and r256["bytes_per_cell"] == 360.0
