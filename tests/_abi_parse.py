"""Parsers for the two sides of the drop-in boundary: the C declarations of include/jrx.h and the Julia mirror structs / ccall
signatures of ext/JustRelaxHIPNativeExt.jl.  Used by tests/test_julia_ext_abi.py (and by scripts/gen_julia_structs.py, which
prints the struct block of the extension from the header)."""
from __future__ import annotations

import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "jrx.h"
JULIA_EXT = ROOT / "ext" / "JustRelaxHIPNativeExt.jl"

C2J = {"double": "Float64", "int32_t": "Int32", "int64_t": "Int64", "uint32_t": "UInt32", "uint8_t": "UInt8", "char": "UInt8"}
DEFINES = {"JRX_MAXPHASE": 8, "JRX_UNIQUE_ID_BYTES": 128}


def _strip_c_comments(txt: str) -> str:
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def _dim(tok: str) -> int:
    tok = tok.strip()
    return DEFINES[tok] if tok in DEFINES else int(tok)


def c_structs(path: Path = HEADER) -> dict:
    """struct name -> [(field, ctype, is_pointer, count)] in declaration order"""
    txt = _strip_c_comments(path.read_text())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"(?:const\s+)?(double|int32_t|int64_t|uint32_t|uint8_t)\s+(.*)$", decl)
            assert mm, f"unparsed declaration in {name}: {decl!r}"
            ctype = mm.group(1)
            for d in mm.group(2).split(","):
                d = d.strip()
                ptr = d.startswith("*")
                d = d.lstrip("* ").replace("const ", "")
                dims = [_dim(x) for x in re.findall(r"\[([^\]]+)\]", d)]
                fname = re.match(r"(\w+)", d).group(1)
                count = 1
                for x in dims:
                    count *= x
                fields.append((fname, ctype, ptr, count))
        out[name] = fields
    return out


def julia_structs(path: Path = JULIA_EXT) -> dict:
    """struct name -> [(field, julia element type, is_pointer, count)]"""
    txt = re.sub(r"#[^\n]*", "", path.read_text())
    out = {}
    for m in re.finditer(r"(?:mutable\s+)?struct\s+(Jrx\w+)\s*\n(.*?)\nend", txt, flags=re.S):
        fields = []
        for part in re.split(r"[;\n]", m.group(2)):
            part = part.strip()
            if not part:
                continue
            mm = re.match(r"(\w+)::(.+)$", part)
            assert mm, f"unparsed Julia field in {m.group(1)}: {part!r}"
            fname, ty = mm.group(1), mm.group(2).strip()
            pm = re.match(r"Ptr\{(\w+)\}$", ty)
            tm = re.match(r"NTuple\{\s*(\d+)\s*,\s*(\w+)\s*\}$", ty)
            tp = re.match(r"NTuple\{\s*(\d+)\s*,\s*Ptr\{(\w+)\}\s*\}$", ty)          # array of pointers: const double *a[6]
            if tp:
                fields.append((fname, tp.group(2), True, int(tp.group(1))))
            elif pm:
                fields.append((fname, pm.group(1), True, 1))
            elif tm:
                fields.append((fname, tm.group(2), False, int(tm.group(1))))
            else:
                fields.append((fname, ty, False, 1))
        out[m.group(1)] = fields
    return out


def julia_name(cname: str) -> str:
    """jrx_stokes3d_fields -> JrxStokes3dFields"""
    return "".join(p.capitalize() for p in cname.split("_"))


def c_prototypes(path: Path = HEADER) -> dict:
    """function name -> [julia ccall argument type, ...] expected for each C parameter"""
    txt = _strip_c_comments(path.read_text())
    txt = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    txt = re.sub(r"typedef\s+enum\s+\w+\s*\{.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(jrx_status|int32_t|int64_t|const char \*)\s*(jrx_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        params = " ".join(m.group(3).split())
        args = []
        if params and params != "void":
            depth, cur, parts = 0, "", []
            for ch in params:
                if ch == "(":
                    depth += 1
                if ch == ")":
                    depth -= 1
                if ch == "," and depth == 0:
                    parts.append(cur)
                    cur = ""
                else:
                    cur += ch
            parts.append(cur)
            for prm in parts:
                args.append(_c_param_to_julia(prm.strip()))
        out[m.group(2)] = args
    return out


def _c_param_to_julia(prm: str) -> str:
    p = prm.replace("const ", "").strip()
    if re.match(r"jrx_handle \*\*", p):
        return "Ref{Ptr{Cvoid}}"
    if re.match(r"jrx_handle \*", p):
        return "Ptr{Cvoid}"
    m = re.match(r"(jrx_\w+) \*", p)
    if m:
        return f"Ref{{{julia_name(m.group(1))}}}"
    if p.startswith("char *"):
        return "Cstring"
    if p.startswith("void *") or re.match(r"jrx_\w+_fn\b", p):      # an opaque context pointer; a callback (typedef ... (*jrx_xxx_fn)(...): @cfunction pointer)
        return "Ptr{Cvoid}"
    m = re.match(r"(double|int32_t|int64_t|uint32_t|uint8_t)\s*(.*)$", p)
    assert m, f"unparsed parameter {prm!r}"
    base, rest = C2J[m.group(1)], m.group(2).strip()
    if rest.startswith("(*"):                       # const int64_t (*ext)[3]
        return f"Ptr{{{base}}}"
    stars = len(re.match(r"^[\s\*]*", rest.replace("const", "")).group(0).replace(" ", ""))
    arr = "[" in rest
    if "*const *" in prm.replace(" *const *", "*const *") or re.search(r"\*\s*const\s*\*", prm):
        return f"Ptr{{Ptr{{{base}}}}}"
    if stars >= 1 and arr:                         # double *const U[3]
        return f"Ptr{{Ptr{{{base}}}}}"
    if stars == 2:
        return f"Ptr{{Ptr{{{base}}}}}"
    if stars == 1 or arr:
        return f"Ptr{{{base}}}"
    return base


def julia_ccalls(path: Path = JULIA_EXT) -> dict:
    """function name -> list of argument-type lists, one per ccall site"""
    txt = re.sub(r"#[^\n]*", "", path.read_text())
    out = {}
    for m in re.finditer(r"ccall\(\s*\(\s*:(jrx_\w+)\s*,\s*libjrx\s*\)\s*,\s*(\w+)\s*,\s*\(", txt):
        i, depth = m.end(), 1
        j = i
        while depth:
            c = txt[j]
            depth += c == "("
            depth -= c == ")"
            j += 1
        inner = txt[i:j - 1]
        parts, cur, d = [], "", 0
        for ch in inner:
            if ch in "{(":
                d += 1
            if ch in "})":
                d -= 1
            if ch == "," and d == 0:
                parts.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur.strip())
        out.setdefault(m.group(1), []).append((m.group(2), [" ".join(p.split()) for p in parts]))
    return out


def same_arg(expected: str, got: str) -> bool:
    """Ref{T} and Ptr{T} are interchangeable at a ccall boundary; Cint is Int32"""
    norm = lambda s: s.replace("Ref{", "Ptr{").replace("Cint", "Int32").replace(" ", "")
    return norm(expected) == norm(got)
