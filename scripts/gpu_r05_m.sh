#!/bin/bash
mkdir -p gpurun_out/r05m
for i in 1 2; do timeout 600 ./scripts/kbench_va 512 6 64 5 2>&1 | grep -E "ms" | cut -c1-110; echo; done | tee gpurun_out/r05m/packed.txt
