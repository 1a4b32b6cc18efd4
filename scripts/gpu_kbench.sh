#!/bin/bash
# kernel micro-benchmark on the GPU box: bash scripts/gpu_kbench.sh <tag> [n] [lines]   (build scripts/kbench first, see the header of kbench.hip)
OUT=gpurun_out/${1:-kbench}
mkdir -p $OUT
timeout 600 scripts/kbench ${2:-512} 20 > $OUT/kbench.log 2>&1
grep -i "fused\|stress\|velocity\|stream" $OUT/kbench.log | tail -${3:-40}
