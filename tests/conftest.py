import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


IPC_WORKERS = {"procs": [], "out": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """`-m gpu` sessions: start the two rank processes of tests/test_gpu_ipc_two_processes.py NOW, before this process makes its first GPU call -- a process
    that has initialised the GPU may not start children on the GPU boxes.  They run beside the other tests (small blocks) and leave their results in a
    temporary directory; the test only reads them."""
    import os
    import socket
    import subprocess
    import tempfile
    expr = session.config.getoption("-m") or ""
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("JRX_NO_IPC_WORKERS"):
        return
    out = tempfile.mkdtemp(prefix="jrx_ipc_")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env0 = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), JRX_IPC_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
    IPC_WORKERS["out"] = out
    for r in range(2):
        log = open(os.path.join(out, f"rank{r}.log"), "w")
        IPC_WORKERS["procs"].append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "_ipc_worker.py")], env=dict(env0, RANK=str(r), LOCAL_RANK=str(r)),
                                                     stdout=log, stderr=subprocess.STDOUT))


def pytest_sessionfinish(session, exitstatus):
    for p in IPC_WORKERS["procs"]:          # only the exact processes started above
        if p.poll() is None:
            p.kill()


@pytest.fixture(scope="session")
def jr():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as orc
    orc.lib()
    return orc
