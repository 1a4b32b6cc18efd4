#!/bin/bash
# final evidence of round 4: GPU suite, profile (kernel stats + PMC + default bench line), the N > 1 bench control flow on one device
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04final}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
bash scripts/gpu_r04_profile.sh ${1:-r04final}
cd $GRAFT_REPO_ROOT
for cfg in "2 512" "8 192"; do
  set -- $cfg
  timeout 900 python3 bench.py --gpus $1 --same-device --default-transport ipc --n $2 --steps 20 --warmup 5 --leg-steps 30 > $OUT/bench_same_device_n$1.json 2> $OUT/bench_same_device_n$1.err
  echo "N=$1 rc=$?"
done
