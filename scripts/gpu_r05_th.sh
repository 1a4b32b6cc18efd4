#!/bin/bash
mkdir -p gpurun_out/r05th
python -m pytest tests/test_gpu_thermal3d.py tests/test_gpu_thermal_multiphase.py tests/test_gpu_field_alloc.py tests/test_gpu_coupled_step.py -m gpu -q > gpurun_out/r05th/tests.log 2>&1
grep -E "passed|failed|^FAILED" gpurun_out/r05th/tests.log | tail -5 | cut -c1-200
for rep in 1 2 3; do for b in 0 1; do python scripts/bench3d_extra.py 0 256 thermal_batch=$b 2>/dev/null | tail -1 | cut -c1-200; done; done
python scripts/bench3d_extra.py 0 384 thermal_batch=0 2>/dev/null | tail -1 | cut -c1-200; python scripts/bench3d_extra.py 0 384 thermal_batch=1 2>/dev/null | tail -1 | cut -c1-200
python scripts/bench3d_extra.py 0 128 thermal_batch=0 2>/dev/null | tail -1 | cut -c1-200; python scripts/bench3d_extra.py 0 128 thermal_batch=1 2>/dev/null | tail -1 | cut -c1-200
