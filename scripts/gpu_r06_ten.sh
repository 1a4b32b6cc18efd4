#!/bin/bash
# ten fresh processes of the driver's bench command (no CPU baseline): k_fused3d launch time on pool-placed arrays and, in the same process, on hipMalloc arrays
tag=${1:-r06ten}; n=${2:-10}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --details $out/d$i.json > $out/b$i.json 2> $out/b$i.err
  python - $out/b$i.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"process: {d['value']:.1f} it/s (20 steps)  {d['steady_state']['value']:.1f} (100 steps)  k_fused3d pool {r['avg_launch_ms']:.3f} ms  hipMalloc {r.get('avg_launch_ms_hipmalloc_arrays') or 0:.3f} ms  general {r['general_form']['avg_launch_ms']:.3f} ms  state_ok {d.get('state_ok')}")
P
done | tee $out/summary.txt
