#!/bin/bash
mkdir -p gpurun_out/r05i
timeout 900 ./scripts/kbench_va 512 6 64 3 > gpurun_out/r05i/va_64.txt 2>&1
grep -v "^rnd" gpurun_out/r05i/va_64.txt | cut -c1-120
timeout 600 ./scripts/kbench_va 512 6 1024 0 > gpurun_out/r05i/va_1024.txt 2>&1
cat gpurun_out/r05i/va_1024.txt | cut -c1-120
