#!/bin/bash
# kernel timeline of one driver-style bench call (20 steps): what runs inside the timed region besides the 19 fused launches?
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04tr}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-general-kernel --no-steady-state --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last 60 kernels: the timed call and what follows
prev_end = None
for r in rows[-75:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e6:10.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us  {r['Kernel_Name'].replace('(anonymous namespace)::', '')[:100]}")
    prev_end = e
PY
rm -rf $OUT/tr
tail -75 $OUT/timeline.txt | cut -c1-170
