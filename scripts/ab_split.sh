#!/bin/bash
# A/B of the forked fused pipeline (option fused_split) on one box: 512^3 and 256^3, alternating
for rep in 1 2; do
  for n in 512 256; do
    for opt in 1 0; do
      echo "n=$n fused_split=$opt"
      python bench.py --n $n --steps 200 --warmup 10 --no-extras --no-cpu-baseline --option fused_split=$opt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('  it/s %.1f  ms/step %.4f  kernel ms %.4f frac %.4f  group ms %.4f  whole frac %.4f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['launch_group_ms'], r['whole_iteration']['frac']))"
    done
  done
done
