#!/bin/bash
mkdir -p gpurun_out/r04c
timeout 1200 python -m pytest tests/test_gpu_two_blocks.py -q -x -m gpu -k "vep3d" > gpurun_out/r04c/pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r04c/pytest.txt | tail -4
timeout 900 python - <<'PY' 2>&1 | grep -v "^iter" | tail -5 | tee gpurun_out/r04c/vep_two_blocks.txt
import sys, json
sys.path.insert(0, ".")
from __graft_entry__ import load_package
jr = load_package()
import bench
r = bench.cfg_multi_rank_path(jr, only=("vep", "z"))
print(json.dumps(r))
PY
