#!/bin/bash
# Round-5 evidence run (one gpurun call): clock probe, the full GPU suite, smoke, the driver's bench command (twice), rocprofv3 kernel stats of the same command, the HBM-side traffic of
# its dominant kernel from separate --pmc passes (python3 directly behind `--`), kernel stats of the 3D VEP leg.   bash scripts/gpu_r05_evidence.sh [tag]
T=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$T
mkdir -p $OUT
bash scripts/clock_probe.sh 4 $T > $OUT/clock_probe.txt 2>&1; grep "clock probe" $OUT/clock_probe.txt | cut -c1-330
python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; grep -E "passed|failed|^FAILED" $OUT/gpu_suite.log | tail -4
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_style_2.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_profiled_run.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 20 --warmup 2 > $OUT/$C.json 2> $OUT/$C.err
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vepstats -- python3 $GRAFT_REPO_ROOT/scripts/bench3d_extra.py 256 0 > $OUT/vep3d_profiled_run.txt 2> $OUT/vepstats.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/bench_kernel_stats.csv
grep -v "at::native\|rocclr" $f | cut -c1-200 | head -7
f=$(find $OUT/vepstats -name "*kernel_stats.csv" | head -1); cp $f $OUT/vep3d_256_kernel_stats.csv
grep -v "at::native\|rocclr" $f | cut -c1-160 | head -8
python3 scripts/pmc_traffic.py $OUT $OUT/pmc_bench_traffic.txt --json $OUT/pmc_traffic.json --source profiles/${T}_pmc_bench_traffic.txt | cut -c1-230
rm -rf $OUT/stats $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/vepstats
python - <<PY
import json
for f in ("bench_driver_style.json", "bench_driver_style_2.json", "bench_profiled_run.json"):
    try:
        d = json.load(open("$OUT/" + f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f, "value %.1f steady %s kernel %.3f ms frac %.3f needed %.3f general %s state %s" % (d["value"], d["steady_state"]["value"] if d.get("steady_state") else None, r["avg_launch_ms"], r["frac"], r.get("frac_at_needed_bytes", 0),
          {k: round(v, 3) for k, v in (r.get("general_form") or {}).items() if isinstance(v, float)}, r.get("device_state")))
PY
