#!/bin/bash
# round 6: the vectorised operand pass: its tests, its kernel time (rocprofv3), the driver's bench command twice
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06r}; mkdir -p $O
python -m pytest tests/test_gpu_stokes3d.py -q -x -k "operand or falls_back or body_forces" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --details $O/d$i.json 2> $O/b$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench:', round(d['value'],1), 'it/s (20)', round(d['steady_state']['value'],1), '(100) kernel', round(r['avg_launch_ms'],3), 'state_ok', d['state_ok'])"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-state-check --no-general-kernel --no-steady-state --details $GRAFT_REPO_ROOT/$O/dp.json > /dev/null 2> $GRAFT_REPO_ROOT/$O/p.err
grep "k_visc_operands_ok" $GRAFT_REPO_ROOT/$O/tr/*/*kernel_stats.csv | cut -c1-60,300-420
