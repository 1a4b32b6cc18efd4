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
