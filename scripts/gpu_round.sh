#!/bin/bash
# One gpurun call: GPU parity tests, a short bench, and a rocprofv3 kernel trace of the same bench command.
# usage (from the repo root on the GPU box): bash scripts/gpu_round.sh [tag] [n]
TAG=${1:-r01}
N=${2:-512}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0)); import os; print('cores', os.cpu_count())" > $OUT/env.log 2>&1
free -g >> $OUT/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 50 --warmup 5 --n 256 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
cat $OUT/bench_256.json
timeout 1200 python bench.py --steps 50 --warmup 5 --n $N > $OUT/bench_$N.json 2> $OUT/bench_$N.err
cat $OUT/bench_$N.json
tail -3 $OUT/bench_$N.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --n $N --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
grep -o '{"metric".*' $OUT/prof.log | tail -1 > $OUT/bench_${N}_profiled_run.json     # the bench line of the profiled process itself
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12
