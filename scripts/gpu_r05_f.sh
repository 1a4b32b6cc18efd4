#!/bin/bash
# round 5, call F: (1) tolerance-mode two-iterations-per-launch kernel; (2) is the kernel's rate a property of the process or of each set of allocations?
mkdir -p gpurun_out/r05f
timeout 600 ./scripts/kbench_x2t 512 10 24 > gpurun_out/r05f/x2t_512.txt 2>&1; cat gpurun_out/r05f/x2t_512.txt | cut -c1-220
timeout 300 ./scripts/kbench_x2t 256 30 24 > gpurun_out/r05f/x2t_256.txt 2>&1; grep -E "ms per launch|worst" gpurun_out/r05f/x2t_256.txt | cut -c1-220
for m in torch 0 1; do for r in 1 2; do timeout 300 python scripts/probe_reroll.py $m 512 4 64 2>&1 | tail -1; done; done | tee gpurun_out/r05f/reroll.txt
