#!/usr/bin/env python3
"""Prints the Julia mirror structs of every struct of include/jrx.h (the block between the GENERATED markers of
ext/JustRelaxHIPNativeExt.jl).  tests/test_julia_ext_abi.py checks, with its own parser, that the extension's structs agree with the header
field by field -- so a header change without regenerating this block fails the CPU suite."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from _abi_parse import C2J, c_structs, julia_name

MUTABLE = {"jrx_solve_result"}
for cname, fields in c_structs().items():
    print(("mutable " if cname in MUTABLE else "") + f"struct {julia_name(cname)}")
    for name, ctype, ptr, count in fields:
        jt = C2J[ctype]
        ty = f"Ptr{{{jt}}}" if ptr and count == 1 else (f"NTuple{{{count}, Ptr{{{jt}}}}}" if ptr else (f"NTuple{{{count}, {jt}}}" if count > 1 else jt))
        print(f"    {name}::{ty}")
    print("end\n")
