#!/bin/bash
# round 6: share of the interior z chunks launched beside the exchange ("fused_first_pct") on a periodic self-neighbour through RCCL, 512^3
out=gpurun_out/${1:-r06p}; mkdir -p $out
for i in 1 2; do for d in x xyz; do for v in 15 22 30 40; do
  python bench.py --gpus 1 --self-halo $d --steps 40 --warmup 5 --no-cpu-baseline --no-state-check --no-general-kernel --option fused_first_pct=$v --details $out/d.json 2> $out/b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('self-halo $d fused_first_pct $v:', round(d['value'],1), 'it/s (40)', round(d['steady_state']['value'],1), '(100)  launch group', round(r.get('avg_launch_ms') or 0,3), 'ms')"
done; done; done | tee $out/summary.txt
