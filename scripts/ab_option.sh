#!/bin/bash
# A/B of a library option on one box: bench.py at the given sizes, alternating.  usage: ab_option.sh OPTION "v1 v2 ..." [reps] ["sizes"]
opt=${1:-fused_xface}; vals=${2:-"0 1"}; reps=${3:-2}; sizes=${4:-"512 256"}
for rep in $(seq $reps); do
  for n in $sizes; do
    for v in $vals; do
      echo -n "n=$n $opt=$v "
      python bench.py --n $n --steps 200 --warmup 10 --no-extras --no-cpu-baseline --option $opt=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(' it/s %.1f  ms/step %.4f  kernel ms %.4f frac %.4f  group ms %.4f  whole frac %.4f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['launch_group_ms'], r['whole_iteration']['frac']))"
    done
  done
done
