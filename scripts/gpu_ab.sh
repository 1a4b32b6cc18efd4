#!/bin/bash
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
for n in 64 96 160 192; do for v in 2 3; do
timeout 600 python bench.py --steps 200 --warmup 10 --n $n --no-cpu-baseline --variant $v > $OUT/b_${n}_v$v.json 2> $OUT/b_${n}_v$v.err
python -c "
import json;d=json.load(open('$OUT/b_${n}_v$v.json'));print('n=$n variant=$v  %9.1f it/s  %8.3f ms/it'%(d['value'],d['ms_per_step']))"
done; done | tee $OUT/ab.txt
