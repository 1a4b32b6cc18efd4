#!/bin/bash
# round 5, call B: which tile order / shape of the headline kernel is least sensitive to the physical backing? (kbench_place: hipMalloc, contiguous, shuffled 2 MiB chunks)
mkdir -p gpurun_out/r05b
for rep in 1 2; do
  for mode in 0 2 1; do
    timeout 300 ./scripts/kbench_place 512 10 $mode > gpurun_out/r05b/kbench_mode${mode}_rep${rep}.txt 2>&1
    cat gpurun_out/r05b/kbench_mode${mode}_rep${rep}.txt
  done
done
python scripts/probe_placement.py torch 512 2>&1 | tail -1
python scripts/probe_placement.py torch 512 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|mall|EA0?_RDREQ|DRAM|TCC_MISS|TCC_HIT|TCC_EA" | head -80 > $GRAFT_REPO_ROOT/gpurun_out/r05b/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r05b/counters.txt
