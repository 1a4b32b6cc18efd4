"""Builds libjrx_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python justrelax.jl_amd/build.py [--force]

-ffp-contract=off: fused multiply-adds appear only where the source says fma(), which is where the
reference has fma/muladd, so device results track the CPU restatement to the last bits.

The library carries a build id = sha256 over csrc/*, include/*.h, the compiler flags, the source list and the ROCm release of the compiler (jrx_build_id()).  A rebuild is skipped only when the existing
.so carries the id of the current sources; `_lib.load()` compares the two again at load time and refuses a stale binary.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "lib" / "libjrx_hip.so"
SRCS = ["handle.hip", "fieldpool.hip", "halo.hip", "stokes3d.hip", "stokes3d_vep.hip", "stokes2d.hip", "thermal2d.hip", "thermal3d.hip", "gridops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-I", str(HERE.parent / "include"), "-I", str(CSRC), "-I", "/opt/rocm/include", "-Wall", "-Wno-unused-function", "-Wno-array-bounds"]
MARKER = b"JRX_BUILD_ID="


def toolchain_version() -> str:
    """the ROCm release the compiler belongs to, read from /opt/rocm/.info/version -- a file, not `hipcc --version`: source_id() also runs inside
    processes that have initialised the GPU (the binding checks the id at load time), and starting a child process from such a process is refused on the
    GPU boxes"""
    rocm = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
    try:
        return (rocm / ".info" / "version").read_text().strip()
    except OSError:
        return ""


def source_id() -> str:
    """sha256 over everything the binary depends on: the sources (file names and contents, sorted), both headers, the compiler flags
    (-ffp-contract=off / -fno-fast-math are what the bit-for-bit agreement with the oracle rests on), the source list and the ROCm release"""
    hsh = hashlib.sha256()
    files = sorted(p for p in CSRC.iterdir() if p.suffix in (".hip", ".hpp")) + sorted((HERE.parent / "include").glob("*.h"))
    for p in files:
        hsh.update(p.name.encode() + b"\0" + p.read_bytes() + b"\0")
    flags = [f for f in FLAGS if not f.startswith("/")]          # paths differ between the build container and the GPU box
    hsh.update(("\0".join(flags) + "\1" + "\0".join(SRCS) + "\1" + toolchain_version()).encode())
    return hsh.hexdigest()


def binary_id(path: Path = OUT):
    """the build id embedded in an existing .so (None if absent), read from the file without loading it"""
    if not path.exists():
        return None
    data = path.read_bytes()
    i = data.find(MARKER)
    if i < 0:
        return None
    j = data.find(b"\0", i)
    return data[i + len(MARKER):j].decode(errors="replace")


def needs_build() -> bool:
    return binary_id() != source_id()


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OUT.parent.mkdir(exist_ok=True)
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)
    sid = source_id()
    procs = []
    for s in SRCS:
        o = objdir / (s + ".o")
        extra = [f'-DJRX_BUILD_ID="{sid}"'] if s == "handle.hip" else []
        cmd = [hipcc, *FLAGS, *extra, "-c", str(CSRC / s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT)] + [str(objdir / (s + ".o")) for s in SRCS] + ["-ldl", "-lrt", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if binary_id() != sid:
        raise RuntimeError("the built library does not carry the build id of its sources")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT, "build id", binary_id())
