#!/bin/bash
mkdir -p gpurun_out/r04pt
for r in 1 2 3; do for f in 0 1; do echo "vep3_prec_tile=$f"; timeout 300 python scripts/bench3d_extra.py 256 0 vep3_prec_tile=$f 2>&1 | grep it_per_s | cut -c1-140; done; done | tee gpurun_out/r04pt/ab.txt
timeout 600 python -m pytest tests/test_gpu_vep3d.py -q -x -m gpu -k "fused_pre_centre or multi_tile" 2>&1 | grep -E "passed|failed" 
bash scripts/pmc_traffic_extra.sh r04pt/pmc 256 0 vep3_prec_tile=1 2>&1 | grep "k_vep3_prec\|options"
