#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04prefrag}
mkdir -p $OUT
for r in 1 2 3 4; do
  for c in 64 0 8 0; do
    timeout 300 python3 scripts/probe_prefrag.py $c > $OUT/p.txt 2> $OUT/p.err
    echo "round $r: $(cat $OUT/p.txt) $(grep -i "error\|Traceback" $OUT/p.err | head -2)"
  done
done
