#!/bin/bash
mkdir -p gpurun_out/r05x
for m in 0 1 0 1; do ./scripts/kbench_loop 512 3 $m > gpurun_out/r05x/loop_$m.txt 2>&1; python3 - gpurun_out/r05x/loop_$m.txt $m <<'PY'
import sys
v = sorted(float(l.split()[1]) for l in open(sys.argv[1]) if l[0].isdigit())
print("operands", "constants" if sys.argv[2] == "1" else "random   ", "k_fused3d<64,8,8> median", v[len(v) // 2], "ms of", len(v), "batches")
PY
done | tee gpurun_out/r05x/data.txt
python scripts/probe_placement.py torch 512 2>&1 | tail -1 | cut -c1-140 | tee -a gpurun_out/r05x/data.txt
