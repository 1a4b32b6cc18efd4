#!/bin/bash
# sample clocks / power / temperature while the headline leg runs (is the run-to-run spread thermal or power capping?)
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower --showtemp --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ' ' | sed 's/GPU\[0\]\s*: //g; s/  */ /g'; echo; sleep 0.7; done ) > gpurun_out/clockwatch.txt &
WATCH=$!
python3 bench.py --no-extras --no-cpu-baseline --no-steady-state --steps 400 --warmup 6 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value %.1f  kernel %.3f ms  general %.1f it/s %.3f ms' % (d['value'], d['roofline']['avg_launch_ms'], d['general_kernel']['it_per_s'], d['general_kernel']['roofline']['avg_launch_ms']))"
wait $WATCH
cat gpurun_out/clockwatch.txt | cut -c1-400
