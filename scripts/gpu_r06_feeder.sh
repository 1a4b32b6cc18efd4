#!/bin/bash
# round 6: the low-x-face feeder lane of the in-kernel neighbour faces (tuning switch nbr_feeder): parity tests, then A/B on a periodic self-neighbour through RCCL (both x faces
# have a neighbour) at 512^3, fresh processes alternating
out=gpurun_out/${1:-r06n}; mkdir -p $out
python -m pytest tests/test_gpu_two_blocks.py tests/test_gpu_ipc_two_processes.py -q -x -k "inkernel or tall or ipc or eight or 2x2x2 or three" 2>&1 | tail -3
for i in 1 2 3; do for v in 0 1; do for d in x xyz; do
  python bench.py --gpus 1 --self-halo $d --steps 40 --warmup 5 --no-cpu-baseline --no-state-check --no-general-kernel --option nbr_feeder=$v --details $out/d.json 2> $out/b.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('self-halo $d nbr_feeder $v:', round(d['value'],1), 'it/s (40)', round(d['steady_state']['value'],1), '(100)  launch group', round(r.get('avg_launch_ms') or 0,3), 'ms')"
done; done; done | tee $out/summary.txt
