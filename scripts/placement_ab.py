#!/usr/bin/env python3
"""Runs scripts/probe_placement.py as a sequence of fresh processes (this driver never touches the GPU) and prints their lines: the A/B of the placement kinds
(VERDICT r4 item 1).  The 2 MiB / 16 MiB chunk legs only run at 512^3 when a 128^3 trial shows that creating and mapping their chunks is affordable."""
import re
import subprocess
import sys
import time
from pathlib import Path
HERE = Path(__file__).resolve().parent
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1100.0
t_start = time.time()


def run(*args, timeout=400):
    if time.time() - t_start > budget_s:
        print("# time budget of this A/B spent; skipped", args, flush=True)
        return ""
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, str(HERE / "probe_placement.py"), *map(str, args)], capture_output=True, text=True, timeout=timeout)
        out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else f"(no output; rc {r.returncode}; {r.stderr.strip()[-300:]})"
    except subprocess.TimeoutExpired:
        out = f"(timed out after {timeout} s)"
    print(f"{out}   [{time.time() - t0:.0f} s]", flush=True)
    return out


def per_chunk_ms(line):
    m = re.search(r"(\d+) chunks created in (\d+) ms, mapped in (\d+) ms", line)
    return (float(m.group(2)) + float(m.group(3))) / max(1.0, float(m.group(1))) if m else 1e9


print(f"# placement A/B at n = {n}", flush=True)
run("torch", n)
run(0, n)
run(2, n)
run(1, n, 64)
cost2 = per_chunk_ms(run(1, 128, 2))
cost16 = per_chunk_ms(run(1, 128, 16))
print(f"# per chunk (create + map): 2 MiB {cost2:.2f} ms, 16 MiB {cost16:.2f} ms", flush=True)
gib = 60.0 * (n / 512.0) ** 3
legs = [("torch",), (0,), (1, 64), (1, 256), (2,), (1, 64, 0, 0, 0), (1, 64, 0, 64, 1), (1, 64, 60000)]
if cost16 * gib * 64 < 90e3:
    legs += [(1, 16)]
if cost2 * gib * 512 < 120e3:
    legs += [(1, 2)]
for rep in range(2):
    for leg in legs:
        run(leg[0], n, *leg[1:])
