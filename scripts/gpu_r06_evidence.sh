#!/bin/bash
# Round-6 evidence run (one gpurun call): the GPU suite as the driver runs it, smoke, the driver's bench command (twice), rocprofv3 kernel stats of the same command, the HBM-side
# traffic of the dominant kernels (separate --pmc passes, MI355X_MICROARCH.md), ten fresh bench processes.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r06_evidence.sh r06z'
tag=${1:-r06z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for i in 1 2; do
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 --details $OUT/bench${i}_details.json ) > $OUT/bench$i.json 2> $OUT/bench$i.err
  wc -c $OUT/bench$i.json
done
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --details $OUT/profiled_details.json > $OUT/bench_profiled_run.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-steady-state --no-state-check --steps 20 --warmup 2 --details $OUT/$C.details.json > $OUT/$C.json 2> $OUT/$C.err
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_traffic.py $OUT $OUT/pmc_bench_traffic.txt --json $OUT/pmc_traffic.json --source profiles/${tag}_pmc_bench_traffic.txt | cut -c1-250
f=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-300
bash scripts/gpu_r06_ten.sh ${tag}_ten 10
