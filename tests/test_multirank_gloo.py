"""N > 1 path on CPU: world_size-2 and world_size-8 (2 x 2 x 2) gloo runs of the decomposition + halo-exchange logic (see _gloo_worker.py)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_halo_and_decomposition(world):
    """2 ranks: (2,1,1); 8 ranks: (2,2,2), the decomposition of BASELINE configs[3] -- edge and corner ghosts through x -> y -> z, the norms' double-counted overlap in all three dimensions"""
    env = dict(os.environ, OMP_NUM_THREADS="1" if world == 8 else "2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "_gloo_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in range(world):
        assert f"rank {k} ok" in r.stdout
