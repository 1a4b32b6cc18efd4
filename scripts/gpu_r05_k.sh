#!/bin/bash
mkdir -p gpurun_out/r05k
for i in 1 2 3; do timeout 300 ./scripts/kbench_realloc 512 8 0 2>&1 | grep -E "round" | cut -c1-200; echo; done | tee gpurun_out/r05k/realloc.txt
