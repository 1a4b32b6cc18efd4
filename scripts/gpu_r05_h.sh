#!/bin/bash
# round 5, call H: the arrays in ONE reserved virtual range, `gap` MiB apart: is the kernel's rate a function of the virtual layout?
mkdir -p gpurun_out/r05h
run() { timeout 200 python scripts/probe_placement.py "$@" 2>&1 | tail -1 | sed 's/ batch.*shuffle 1//; s/setup.*//' ; }
( for rep in 1 2; do
  for gap in 0 2 6 14 30 62 126 254 510 1022 18 50 98 200; do run 1 512 64 0 0 1 25 256 $gap; done
done
for gap in 0 2 6 62 126 510; do run 1 512 1024 0 0 1 25 256 $gap; done
for al in 64 1024; do run 1 512 64 0 $al 1 25 256 0; run 1 512 64 0 $al 1 25 256 62; done
run 1 512 64 0 0 1 25 0 0; run 1 512 64 0 0 1 25 0 0; run torch 512; run torch 512 ) | tee gpurun_out/r05h/arena_gap.txt
