#!/bin/bash
# headline bench over a range of local grid sizes (one GPU): it/s, ms/iteration, roofline fraction at 360 B/cell
OUT=gpurun_out/${1:-sizes}
mkdir -p $OUT
for n in 64 96 128 192 256 320 384 448 496 512 576; do
timeout 600 python bench.py --steps 100 --warmup 10 --n $n --no-cpu-baseline > $OUT/bench_$n.json 2> $OUT/bench_$n.err
python -c "
import json;d=json.load(open('$OUT/bench_$n.json'));r=d['roofline'];print('n=%4d  %9.1f it/s  %8.3f ms/it  whole-iteration frac %.3f  kernel: %s'%($n,d['value'],d['ms_per_step'],r.get('whole_iteration',{}).get('frac',r.get('frac',0)),r['kernel'][:28]))"
done | tee $OUT/sizes.txt
