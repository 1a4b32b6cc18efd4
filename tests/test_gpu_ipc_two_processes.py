"""The cross-process copy-engine transport (jrx_comm_init_ipc) with one rank per PROCESS, two processes on one device.

The two rank processes (tests/_ipc_worker.py) are started by tests/conftest.py at session start, before this process touches the GPU; each runs its block
of a (2,1,1), (1,1,2) or (1,2,1) decomposition through solve! in all four pipelines (fused kernel with the exchange beside it / behind it / shell tiles,
split sweeps with hidden communication) and requires it to equal the undecomposed device run bit for bit -- what tests/test_gpu_two_blocks.py checks for
the in-process transport.  Reference: update_halo!(V) inside @hide_communication (src/stokes/Stokes3D.jl:104-121), one process per rank
(test/runtests.jl:73-90)."""
import json
import time
from pathlib import Path

import pytest

import conftest

pytestmark = pytest.mark.gpu


def test_two_processes_on_one_device_equal_the_undecomposed_run():
    procs, out = conftest.IPC_WORKERS["procs"], conftest.IPC_WORKERS["out"]
    if not procs:
        pytest.skip("the rank processes are only started by `-m gpu` sessions (tests/conftest.py)")
    t0 = time.time()
    while any(p.poll() is None for p in procs) and time.time() - t0 < 900:
        time.sleep(0.5)
    logs = {r: (Path(out) / f"rank{r}.log").read_text()[-3000:] for r in range(2)}
    assert all(p.poll() is not None for p in procs), ("the rank processes did not finish", logs)
    res = []
    for r in range(2):
        f = Path(out) / f"rank{r}.json"
        assert f.exists(), (r, logs[r])
        res.append(json.loads(f.read_text()))
    for r, d in enumerate(res):
        assert d.get("skipped") is None, d
        assert d["error"] is None, (r, d["error"], logs[r])
        assert d.get("done") and procs[r].returncode == 0, (r, logs[r])
        names = [c["case"] for c in d["cases"]]
        assert names[0] == "update_halo" and len(names) >= 1 + 3 * 4 + 2 * 2, names
        for c in d["cases"]:
            assert c["ok"], (r, c)
            if "fused_launches" in c:
                assert (c["fused_launches"] >= 12) == ("split_sweeps" not in c["case"]), c
        assert d["norm_bits_equal_across_ranks"]
        visc = [c for c in d["cases"] if "inkernel_launches" in c]
        assert len(visc) == 3 and all(c["inkernel_launches"] >= 12 for c in visc), visc        # dt = Inf: the kernel finished the neighbour faces itself
    # both ranks report the same norms, and both dims (2,1,1) and (1,1,2) ran in all four pipelines
    for dims in ("(2, 1, 1)", "(1, 1, 2)"):
        for pipe in ("fused", "fused_overlap", "fused_early", "split_sweeps"):
            assert any(c["case"].startswith(f"dims={dims} n=(70, 13, 12) {pipe}") for c in res[0]["cases"]), (dims, pipe)
