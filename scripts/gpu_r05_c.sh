#!/bin/bash
# round 5, call C: taller tiles of the headline kernel (kbench_place), 512^3 under hipMalloc (two processes) and contiguous backing, and 256^3
mkdir -p gpurun_out/r05c
for tag in "512 10 0 a" "512 10 0 b" "512 10 2 c" "512 10 1 d" "256 40 0 e" "256 40 0 f"; do
  set -- $tag
  timeout 300 ./scripts/kbench_place $1 $2 $3 > gpurun_out/r05c/kbench_n$1_mode$3_$4.txt 2>&1
  echo "== n $1 mode $3 ($4)"; grep " ms " gpurun_out/r05c/kbench_n$1_mode$3_$4.txt | awk '{printf "%s %s %s %s  %s\n", $1, $2, $3, $4, $NF}' | paste - - - | column -t
done
