#!/bin/bash
mkdir -p gpurun_out/r04n
timeout 900 python -m pytest tests/test_gpu_vep3d.py tests/test_gpu_small_grid_graphs.py -q -x -m gpu 2>&1 | tail -3
timeout 900 python scripts/bench_vep3d_switch.py vep3_np_const 16 32 48 64 96 128 256 2>&1 | grep '"n"' | tee gpurun_out/r04n/np_const.txt
timeout 900 python scripts/bench_vep3d_switch.py vep3_fuse_pc 16 32 48 64 96 128 256 2>&1 | grep '"n"' | tee gpurun_out/r04n/fuse_pc.txt
