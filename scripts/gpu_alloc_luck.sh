#!/bin/bash
# run-to-run spread of the headline leg in fresh processes on one box, with and without one big reservation first (bench.py --slab-gb)
cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5; do
  for slab in 0 80; do
    python3 bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 60 --warmup 6 --slab-gb $slab 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slab_gb $slab  value %.1f  kernel %.3f ms  general %.1f it/s %.3f ms' % (d['value'], d['roofline']['avg_launch_ms'], d['general_kernel']['it_per_s'], d['general_kernel']['roofline']['avg_launch_ms']))"
  done
done
