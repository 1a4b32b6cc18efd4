#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04flip2}
mkdir -p $OUT
timeout 600 python3 scripts/bench_end_flips.py 512 20 > $OUT/ab.txt 2> $OUT/ab.err
timeout 600 python3 scripts/bench_end_flips.py 256 20 >> $OUT/ab.txt 2>> $OUT/ab.err
cat $OUT/ab.txt; tail -2 $OUT/ab.err
