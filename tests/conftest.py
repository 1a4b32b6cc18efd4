import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


IPC_WORKERS = {"procs": [], "out": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU gate (`pytest -m gpu -x`): oracle parity first -- the BASELINE sizes, the golden fixtures, the kernel-level comparisons of every path, the halo exchange --
# then the decomposed runs, and LAST everything that tests an allocator, a replay mechanism or a host protocol rather than a result of the reference.  A failure of the second kind must
# not un-prove the first (round 5: one allocator experiment stopped the gate at test 35 of 598).
GPU_ORDER = ["test_gpu_baseline_sizes", "test_gpu_golden", "test_gpu_stokes3d", "test_gpu_stokes2d_thermal", "test_gpu_thermal3d", "test_gpu_vep2d", "test_gpu_vep3d", "test_gpu_halo",
             "test_gpu_bcs", "test_gpu_creep", "test_gpu_coupled_step", "test_gpu_thermal_multiphase", "test_gpu_vep_extras", "test_gpu_gridops", "test_gpu_nonuniform", "test_gpu_fullsize",
             "test_checkpoint_roundtrip", "test_gpu_two_blocks", "test_gpu_ipc_two_processes", "test_gpu_small_grid_graphs", "test_gpu_field_alloc"]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(GPU_ORDER)}
    def key(it):
        mod = Path(str(it.fspath)).stem
        return rank.get(mod, len(GPU_ORDER) - 2.5 if mod.startswith("test_gpu") else -1)       # CPU files keep their place in front; an unlisted GPU file runs before the allocator / replay group
    items.sort(key=key)                                                                        # stable: the order inside a file is kept


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
