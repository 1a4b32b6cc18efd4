#!/bin/bash
# physically contiguous arrays (the slow rate, deterministically) with array starts staggered by k * S bytes: which S brings the fast rate back?
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04contig}
mkdir -p $OUT
for r in 1 2; do
  for S in 0 4352 69888 1118464 16847104 2101504; do
    timeout 300 python3 scripts/probe_contiguous.py 1 512 $S > $OUT/p_${S}_$r.txt 2> $OUT/p_${S}_$r.err
    echo "round $r: $(cat $OUT/p_${S}_$r.txt) $(grep -i "error\|Traceback" $OUT/p_${S}_$r.err | head -2)"
  done
done
timeout 300 python3 scripts/probe_contiguous.py 0 512 0 > $OUT/p_plain.txt 2> $OUT/p_plain.err; echo "plain: $(cat $OUT/p_plain.txt)"
