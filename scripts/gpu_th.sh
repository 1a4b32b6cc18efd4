#!/bin/bash
OUT=gpurun_out/${1:-th}
mkdir -p $OUT
for c in "10404 8" "10404 1" "10404 2" "10404 4" "10402 1" "10402 2" "10408 1" "10404 8"; do
set -- $c
JRX_TH_CFG=$1 JRX_TH_XG=$2 timeout 600 python scripts/bench3d_extra.py 0 256 2>/dev/null | tail -1 | cut -c1-140 | sed "s/^/cfg=$1 xg=$2 /"
done | tee $OUT/th.log
