#!/bin/bash
mkdir -p gpurun_out/r05g
python -m pytest tests/test_gpu_field_alloc.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python scripts/probe_reroll2.py 512 64 6 1 2>&1 | tee gpurun_out/r05g/reroll2_64.txt | cut -c1-1500
timeout 600 python scripts/probe_reroll2.py 512 1024 6 0 2>&1 | tee gpurun_out/r05g/reroll2_1024.txt | cut -c1-600
timeout 600 python scripts/probe_reroll2.py 512 2 4 0 2>&1 | tee gpurun_out/r05g/reroll2_2.txt | cut -c1-600
