#!/bin/bash
# round-4: folded high-face layers (tests + A/B) and the N > 1 bench control flow on one device
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stokes3d.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_sizes.py tests/test_gpu_golden.py tests/test_gpu_two_blocks.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for hf in 1 0; do
  timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-general-kernel --option fused_hiface=$hf > $OUT/bench512_hif$hf.json 2> $OUT/bench512_hif$hf.err
  timeout 600 python3 bench.py --n 256 --steps 400 --warmup 20 --no-extras --no-cpu-baseline --no-general-kernel --option fused_hiface=$hf > $OUT/bench256_hif$hf.json 2> $OUT/bench256_hif$hf.err
done
python3 - <<PY
import json
for n in (512, 256):
    for hf in (1, 0):
        try:
            d = json.load(open("$OUT/bench%d_hif%d.json" % (n, hf)))
            print(n, "hiface", hf, round(d["value"], 1), "it/s  kernel", round(d["roofline"]["avg_launch_ms"], 4), "ms  group", round(d["roofline"]["launch_group_ms"], 4), "whole frac", round(d["roofline"]["whole_iteration"]["frac"], 4))
        except Exception as e:
            print(n, hf, "error", e)
PY
for cfg in "2 384" "4 256" "8 192"; do
  set -- $cfg
  timeout 900 python3 bench.py --gpus $1 --same-device --default-transport ipc --n $2 --steps 20 --warmup 5 --leg-steps 30 > $OUT/bench_same_device_n$1.json 2> $OUT/bench_same_device_n$1.err
  echo "N=$1 rc=$?"; tail -c 400 $OUT/bench_same_device_n$1.json; echo; tail -2 $OUT/bench_same_device_n$1.err | cut -c1-300
done
