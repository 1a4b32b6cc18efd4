#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04stg}
mkdir -p $OUT
timeout 1500 python3 scripts/bench_alloc_stagger.py 512 0 4352 69888 1118464 > $OUT/stagger.txt 2> $OUT/stagger.err
cat $OUT/stagger.txt | cut -c1-150; tail -3 $OUT/stagger.err
