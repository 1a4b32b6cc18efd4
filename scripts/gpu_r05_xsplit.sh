#!/bin/bash
# the x split of the two-block leg under tuning switches (one process per setting; overhead = median of five alternations on the same handles)
for O in "" "fused_first_pct=5" "fused_first_pct=30" "fused_first_pct=50" "fused_first_pct=80"; do
  python3 scripts/bench_multi_rank.py 512 40 x default $O 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)['block_it_per_s']['default']
print('x', '$O' or 'defaults', {k:round(v,2) for k,v in d.items()})"
done
