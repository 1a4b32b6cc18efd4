#!/bin/bash
# kbench only
OUT=gpurun_out/${1:-ylds}
mkdir -p $OUT
timeout 600 scripts/kbench ${2:-512} 20 > $OUT/kbench.log 2>&1
grep -i "fused" $OUT/kbench.log | tail -${3:-30}
