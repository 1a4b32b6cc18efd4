#!/bin/bash
OUT=gpurun_out/${1:-vep}
mkdir -p $OUT
for m in 1 2; do
timeout 600 python scripts/bench3d_extra.py 256 0 2>/dev/null | tail -1 | cut -c1-200
done | tee $OUT/vep.log
timeout 600 python scripts/bench3d_extra.py 256 256 2>/dev/null | tail -2 | cut -c1-200 | tee -a $OUT/vep.log
