"""CPU sanitizer job for the host side (CPU builds only -- no GPU sanitizers on this pool).

* The host protocol of the two copy-engine transports (csrc/ipc_ctl.hpp: control segment of the ipc transport -- attach, failure flag, waits with a time-out, all-reduce, the
  per-face handshake; csrc/local_group.hpp: the in-process group) is free of HIP, and tests/host/ctl_harness.cpp builds exactly those headers with g++ under
  -fsanitize=thread and -fsanitize=address,undefined: threads and forked processes play 2 .. 8 ranks, including a rank that leaves and a rank that never comes.
* The C restatement under oracle/ (test infrastructure) is built with -fsanitize=address,undefined (`make -C oracle sanitize`) and a slice of its own tests runs on that build.

Recipe by hand:
    g++ -std=c++17 -O1 -g -fsanitize=thread -I justrelax.jl_amd/csrc tests/host/ctl_harness.cpp -o /tmp/ctl_tsan -lpthread -lrt && /tmp/ctl_tsan threads 8 100
    make -C oracle sanitize && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 JRX_ORACLE_LIB=$PWD/oracle/libjrx_oracle_asan.so \
        python -m pytest tests/test_oracle_golden.py -q
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GXX = shutil.which("g++")
pytestmark = pytest.mark.skipif(GXX is None, reason="g++ not found")


def _build(tmp_path, name, flags):
    exe = tmp_path / name
    subprocess.run([GXX, "-std=c++17", "-O1", "-g", *flags, "-I", str(ROOT / "justrelax.jl_amd" / "csrc"), str(ROOT / "tests" / "host" / "ctl_harness.cpp"), "-o", str(exe),
                    "-lpthread", "-lrt"], check=True)
    return exe


def _run(exe, *args, env=None):
    r = subprocess.run([str(exe), *map(str, args)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (args, r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_transport_protocols_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "ctl_tsan", ["-fsanitize=thread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1")
    for mode, n, k in (("threads", 2, 60), ("threads", 5, 60), ("threads", 8, 40), ("local", 2, 100), ("local", 8, 100)):
        _run(exe, mode, n, k, env=env)


def test_transport_protocols_under_address_and_ub_sanitizers(tmp_path):
    exe = _build(tmp_path, "ctl_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    for mode, n, k in (("threads", 4, 40), ("procs", 2, 100), ("procs", 8, 60), ("local", 4, 60)):
        _run(exe, mode, n, k)


def test_a_weakened_flag_protocol_is_reported(tmp_path):
    """the harness must be able to see what it guards: with the release / acquire of the flags taken away ThreadSanitizer reports the payload race"""
    mut = tmp_path / "mut"
    mut.mkdir()
    src = (ROOT / "justrelax.jl_amd" / "csrc" / "ipc_ctl.hpp").read_text()
    weak = src.replace("__atomic_store_n(p, v, __ATOMIC_RELEASE)", "__atomic_store_n(p, v, __ATOMIC_RELAXED)").replace("__atomic_load_n(p, __ATOMIC_ACQUIRE)", "__atomic_load_n(p, __ATOMIC_RELAXED)")
    assert weak != src
    (mut / "ipc_ctl.hpp").write_text(weak)
    shutil.copy(ROOT / "justrelax.jl_amd" / "csrc" / "local_group.hpp", mut / "local_group.hpp")
    exe = tmp_path / "ctl_tsan_mut"
    subprocess.run([GXX, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", str(mut), str(ROOT / "tests" / "host" / "ctl_harness.cpp"), "-o", str(exe), "-lpthread", "-lrt"], check=True)
    r = subprocess.run([str(exe), "threads", "4", "50"], capture_output=True, text=True, timeout=300)
    assert "ThreadSanitizer: data race" in r.stderr


def test_oracle_under_address_and_ub_sanitizers():
    """the C restatement built with -fsanitize=address,undefined runs its known-answer tests (mini kernels, boundary conditions, VEP kernels, material laws, grid operators)"""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not found")
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "sanitize"], check=True, capture_output=True)
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not Path(asan).is_absolute():
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", JRX_ORACLE_LIB=str(ROOT / "oracle" / "libjrx_oracle_asan.so"))
    sel = ["tests/test_oracle_bcs.py", "tests/test_oracle_vep_kernels.py", "tests/test_oracle_material.py", "tests/test_oracle_gridops.py"]
    # the golden file takes ~50 s on the sanitizer build: its stencil / utility known answers and the two 3D diffusion runs always, the whole file when asked for (JRX_SANITIZE_ALL=1)
    if os.environ.get("JRX_SANITIZE_ALL"):
        sel.append("tests/test_oracle_golden.py")
    else:
        sel += ["tests/test_oracle_golden.py::" + t for t in ("test_mini_kernels_accessors", "test_mini_kernels_differences", "test_mini_kernels_averages", "test_mini_kernels_clamped",
                                                                "test_mysum", "test_utils_known_answers", "test_diffusion2d", "test_diffusion3d", "test_thermal_phase_helpers_known_answers")]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *sel], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
