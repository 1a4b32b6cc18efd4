#!/bin/bash
# 3D VEP experiments: parity, then A/B at 256^3 (alternating), then a kernel profile
mkdir -p gpurun_out/r04v
rm -f gpurun_out/r04v/ab_256.txt
timeout 900 python -m pytest tests/test_gpu_vep3d.py tests/test_gpu_small_grid_graphs.py "tests/test_gpu_fullsize.py::test_vep3d_edge_kernel_forms_agree_at_full_size" -q -x -m gpu 2>&1 | tail -15 > gpurun_out/r04v/pytest.txt
cat gpurun_out/r04v/pytest.txt
for r in 1 2 3; do
  for f in "vep3_edges=4" "vep3_edges=6"; do
    echo "$f" >> gpurun_out/r04v/ab_256.txt
    timeout 300 python scripts/bench3d_extra.py 256 0 $f 2>&1 | grep it_per_s | cut -c1-140 >> gpurun_out/r04v/ab_256.txt
  done
done
cat gpurun_out/r04v/ab_256.txt
cd /tmp && export TMPDIR=/tmp
for v in "vep3_edges=4"; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04v/prof_$v -- python3 $GRAFT_REPO_ROOT/scripts/bench3d_extra.py 256 0 $v > $GRAFT_REPO_ROOT/gpurun_out/r04v/prof_$v.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r04v/prof_$v -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r04v/kernel_stats_$v.csv && head -7 "$f" | cut -c1-150,330-420
done
