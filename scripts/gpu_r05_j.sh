#!/bin/bash
mkdir -p gpurun_out/r05j
for i in 1 2 3; do timeout 300 ./scripts/kbench_stream 512 8 0 2>&1 | grep -E "stream|pass" | cut -c1-260; done | tee gpurun_out/r05j/streams.txt
