"""Builds libjrx_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python justrelax.jl_amd/build.py [--force]

-ffp-contract=off: fused multiply-adds appear only where the source says fma(), which is where the
reference has fma/muladd, so device results track the CPU restatement to the last bits.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "lib" / "libjrx_hip.so"
SRCS = ["handle.hip", "halo.hip", "stokes3d.hip", "stokes3d_vep.hip", "stokes2d.hip", "thermal2d.hip", "thermal3d.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-I", str(HERE.parent / "include"), "-I", str(CSRC), "-I", "/opt/rocm/include", "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*")) + [HERE.parent / "include" / "jrx.h", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OUT.parent.mkdir(exist_ok=True)
    objdir = HERE / "build"
    objdir.mkdir(exist_ok=True)
    procs = []
    for s in SRCS:
        o = objdir / (s + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(CSRC / s), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT)] + [str(objdir / (s + ".o")) for s in SRCS] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
