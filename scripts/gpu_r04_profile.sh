#!/bin/bash
# Round-4 evidence run (one gpurun call): the default bench line, rocprofv3 kernel stats of the HEADLINE leg alone, and the HBM-side traffic of its dominant kernel
# from separate --pmc passes (python3 directly behind `--`).   bash scripts/gpu_r04_profile.sh [tag]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-steady-state > $OUT/headline_profiled_run.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 20 --warmup 2 > $OUT/$C.json 2> $OUT/$C.err
done
cd $GRAFT_REPO_ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/bench_headline_kernel_stats.csv
grep -v "at::native\|rocclr" $f | cut -c1-220 | head -8
python3 scripts/pmc_traffic.py $OUT $OUT/pmc_bench_traffic.txt --json $OUT/pmc_traffic.json --source profiles/${1:-r04}_pmc_bench_traffic.txt
rm -rf $OUT/stats $OUT/FETCH_SIZE $OUT/WRITE_SIZE
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json
