#!/bin/bash
mkdir -p gpurun_out/r04n
timeout 900 python -m pytest tests/test_gpu_vep2d.py tests/test_gpu_vep_extras.py tests/test_gpu_small_grid_graphs.py -q -x -m gpu 2>&1 | tail -3
timeout 900 python scripts/bench_vep2d_switch.py vep3_np_const 64 128 256 512 1024 2048 2>&1 | grep '"n"' | tee gpurun_out/r04n/np_const_2d.txt
