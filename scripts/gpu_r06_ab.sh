#!/bin/bash
# A/B of a tuning option on the driver's bench command, alternating fresh processes:  bash scripts/gpu_r06_ab.sh TAG "opt=val" [pairs]
tag=$1; opt=$2; pairs=${3:-2}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $pairs); do
  for v in base opt; do
    extra=""; [ $v = opt ] && extra="--option $opt"
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-state-check $extra --details $out/d_${v}$i.json > $out/b_${v}$i.json 2> $out/b_${v}$i.err
    python - $out/b_${v}$i.json "$v $extra" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"{sys.argv[2]:28s}: {d['value']:.1f} it/s (20)  {d['steady_state']['value']:.1f} (100)  k_fused3d {r['avg_launch_ms']:.3f} ms  general {r['general_form']['avg_launch_ms']:.3f} ms")
P
  done
done | tee $out/summary.txt
