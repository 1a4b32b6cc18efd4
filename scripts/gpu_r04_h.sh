#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 700 python3 scripts/bench_multi_rank_modes.py 512 > $OUT/multi_rank_512.json 2> $OUT/multi_rank_512.err
python3 - <<PY
import json
d = json.loads(open("$OUT/multi_rank_512.json").read().strip().splitlines()[-1])
for s in ("split_x", "split_z"):
    v = d[s]
    print(s, "uncoupled", round(v["two_uncoupled_blocks_block_it_per_s"], 1), {m: (round(v[m]["block_it_per_s"], 1), round(v[m]["overhead_pct"], 2)) for m in ("inkernel", "serial", "early", "overlap")})
ip = d.get("ipc_two_processes", {})
for s in ("split_x", "split_z"):
    if s in ip:
        print("ipc", s, "uncoupled", round(ip[s]["two_uncoupled_blocks_block_it_per_s"], 1), {m: (round(ip[s][m]["block_it_per_s"], 1), round(ip[s][m].get("overhead_pct", 0), 2)) for m in ("inkernel", "early", "serial")})
        print("   chain inkernel", ip[s]["inkernel"]["chain_us_per_rank"])
print(ip.get("error"))
PY
tail -3 $OUT/multi_rank_512.err
