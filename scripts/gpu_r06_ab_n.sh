#!/bin/bash
# A/B of tuning options at a given block size, alternating fresh processes:  bash scripts/gpu_r06_ab_n.sh TAG N PAIRS "opt=val" ["opt2=val2" ...]
tag=$1; n=$2; pairs=$3; shift 3
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $pairs); do
  for v in base "$@"; do
    extra=""; [ "$v" != base ] && extra="--option $v"
    python bench.py --gpus 1 --n $n --steps 40 --warmup 5 --no-cpu-baseline --no-state-check --no-general-kernel $extra --details $out/d.json > $out/b.json 2> $out/b.err
    python - $out/b.json "n $n $v" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"{sys.argv[2]:32s}: {d['value']:.1f} it/s (40)  {d['steady_state']['value']:.1f} (100)  k_fused3d {r['avg_launch_ms']:.4f} ms")
P
  done
done | tee $out/summary.txt
