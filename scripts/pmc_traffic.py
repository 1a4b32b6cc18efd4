#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 20 --warmup 2` (one counter per pass,
as MI355X_MICROARCH.md prescribes) into (i) a readable table and (ii) profiles/pmc_traffic.json, the file bench.py takes `roofline.traffic` from.  The JSON carries the
sha256 of csrc/stokes3d_kernels.hpp and the git commit of the tree the passes ran on, so that bench.py can tell when the figures are stale.

    python3 scripts/pmc_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ subdirectories> <table.txt> [--json profiles/pmc_traffic.json] [--source profiles/<name>.txt]
"""
import collections
import csv
import glob
import hashlib
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    out_dir, table = Path(sys.argv[1]), Path(sys.argv[2])
    jpath = Path(sys.argv[sys.argv.index("--json") + 1]) if "--json" in sys.argv else None
    source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else str(table)
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(str(out_dir / c / "*" / "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                res[r["Kernel_Name"].replace("(anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    n = 512.0 ** 3
    lines = ["# python3 bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 20 --warmup 2 under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each); FETCH_SIZE doubled",
             "# (gfx950: 128-B read requests tallied at 64 B, MI355X_MICROARCH.md 'HBM'); rocprofv3 reports KiB; passes = bytes / (8 B x 512^3)",
             "# k_fused3d<..., TAG, VISC, HIF, VFOLD, NBR, NOF>: VISC true = viscous-limit form (dt = Inf), false = general form (the general_kernel leg); HIF true = high-face layers inside;",
             "# NOF (last argument) 2 = the three zero body-force arrays of SolVi3D are not loaded (the headline), 0 = they are (the with_body_forces leg)"]
    tot = {}
    for k, d in sorted(res.items()):
        if "at::" in k or "rocclr" in k:
            continue
        fv, wv = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        fe = 2.0 * 1024.0 * sum(fv) / max(len(fv), 1)
        wr = 1024.0 * sum(wv) / max(len(wv), 1)
        tot[k] = fe + wr
        lines.append(f"{k[:110]:110s} launches {len(fv):4d}  fetch {fe / 1e9:8.3f} GB ({fe / 8 / n:5.1f} passes)  write {wr / 1e9:8.3f} GB ({wr / 8 / n:5.1f} passes)  total {(fe + wr) / 1e9:8.3f} GB")
    table.write_text("\n".join(lines) + "\n")
    print("\n".join(l for l in lines if "k_fused3d" in l or "k_stress3d_zb" in l))
    if jpath:
        def pick(pred):
            c = [v for k, v in tot.items() if pred(k)]
            return max(c) if c else None
        targs = lambda k: [a.strip() for a in k[k.index("<") + 1:k.index(">")].split(",")] if "<" in k else []
        fused = {k: v for k, v in tot.items() if "k_fused3d<" in k}
        nof = lambda k: int(targs(k)[16]) if len(targs(k)) >= 17 else 0        # template argument NOF: body-force arrays not loaded (0 none, 1 ρg_x, ρg_y, 2 all three)
        visc = pick(lambda k: "k_fused3d<" in k and len(targs(k)) >= 13 and targs(k)[12] == "true" and nof(k) == 0)
        visc_n = {l: pick(lambda k: "k_fused3d<" in k and len(targs(k)) >= 13 and targs(k)[12] == "true" and nof(k) == l) for l in (1, 2)}
        gen = pick(lambda k: "k_fused3d<" in k and (len(targs(k)) < 13 or targs(k)[12] == "false") and nof(k) == 0)
        gen_n = {l: pick(lambda k: "k_fused3d<" in k and len(targs(k)) >= 13 and targs(k)[12] == "false" and nof(k) == l) for l in (1, 2)}
        zb_v = pick(lambda k: "k_stress3d_zb<" in k and targs(k)[-1] == "true")
        zb_g = pick(lambda k: "k_stress3d_zb<" in k and targs(k)[-1] == "false")
        sha = hashlib.sha256((ROOT / "justrelax.jl_amd" / "csrc" / "stokes3d_kernels.hpp").read_bytes()).hexdigest()
        head = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
        json.dump({"what": "L2<->fabric bytes per launch of the dominant kernels at n = 512 from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE doubled per the gfx950 "
                           "correction of MI355X_MICROARCH.md), written by scripts/pmc_traffic.py; bench.py prints `traffic: null` when csrc/stokes3d_kernels.hpp differs from the file these passes ran",
                   "kernels_sha256": sha, "git_head": head, "source": source, "n": 512,
                   "k_fused3d_general": gen, "k_fused3d_visc": visc, "k_fused3d_visc_nof1": visc_n[1], "k_fused3d_visc_nof2": visc_n[2], "k_fused3d_general_nof1": gen_n[1], "k_fused3d_general_nof2": gen_n[2], "k_stress3d_zb_general": zb_g, "k_stress3d_zb_visc": zb_v}, open(jpath, "w"), indent=1)
        print("wrote", jpath, "visc", visc, "visc without body forces", visc_n, "general", gen)


if __name__ == "__main__":
    main()
