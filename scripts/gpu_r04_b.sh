#!/bin/bash
# round-4: temporal-blocking prototype timing + the N > 1 bench control flow on one device (ipc / local_peer transports)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 scripts/kbench_tb 512 20 > $OUT/kbench_tb_512.log 2>&1
cat $OUT/kbench_tb_512.log
timeout 300 scripts/kbench_tb 256 50 > $OUT/kbench_tb_256.log 2>&1
grep -v "^kbench" $OUT/kbench_tb_256.log | head -20
for cfg in "2 384" "4 256" "8 192"; do
  set -- $cfg
  timeout 900 python3 bench.py --gpus $1 --same-device --default-transport ipc --n $2 --steps 20 --warmup 5 --leg-steps 30 > $OUT/bench_same_device_n$1.json 2> $OUT/bench_same_device_n$1.err
  echo "N=$1 rc=$?"; tail -c 600 $OUT/bench_same_device_n$1.json; tail -3 $OUT/bench_same_device_n$1.err
done
