#!/bin/bash
mkdir -p gpurun_out/r05l
for i in 1 2; do timeout 600 ./scripts/kbench_va 512 6 64 4 2>&1 | grep -E "chunk .* MiB, stride|packed" | cut -c1-120; echo; done | tee gpurun_out/r05l/chunks.txt
