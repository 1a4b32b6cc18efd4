#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
cp -r /tmp/jrx_ipc_* $OUT/ 2>/dev/null
timeout 900 python3 bench.py --gpus 2 --same-device --default-transport ipc --n 512 --steps 20 --warmup 5 --leg-steps 30 --option comm_bcs_lazy=1 > $OUT/bench_n2_lazy1.json 2> $OUT/bench_n2_lazy1.err
timeout 900 python3 bench.py --gpus 2 --same-device --default-transport ipc --n 512 --steps 20 --warmup 5 --leg-steps 30 --option comm_bcs_lazy=0 > $OUT/bench_n2_lazy0.json 2> $OUT/bench_n2_lazy0.err
python3 - <<PY
import json
for l in (1, 0):
    try:
        d = json.load(open("$OUT/bench_n2_lazy%d.json" % l))
        t = d["transports"]
        print("lazy", l, "ipc", round(t["ipc"]["it_per_s"], 1), "local_peer", round(t["local_peer"]["it_per_s"], 1), "alt", round(d["alt_decomposition"]["it_per_s"], 1), "chain", t["ipc"]["chain_us_per_rank"])
    except Exception as e:
        print(l, "error", e)
PY
