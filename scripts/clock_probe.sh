#!/bin/bash
# clock_probe.sh [seconds=6] [tag]: the headline kernel (64 x 8 tile) in a loop beside a 5 Hz log of every visible device's sclk / power / temperatures (hwmon); prints one summary line.
# Run at the start of every GPU call of round 5: the spread of the kernel's rate between boxes and processes is read beside the clock the power management sustains under the 1400 W cap.
S=${1:-6}; TAG=${2:-probe}
O=gpurun_out/clock_probes; mkdir -p $O
ID=$(date +%s)
( while true; do
    line="$(date +%s.%N | cut -c1-13)"
    for d in /sys/class/drm/card*/device; do
      h=$(ls -d $d/hwmon/hwmon* 2>/dev/null | head -1)
      [ -n "$h" ] && [ -r $h/freq1_input ] && line="$line | $(cat $h/freq1_input) $(cat $h/power1_input 2>/dev/null) $(cat $h/temp2_input 2>/dev/null) $(cat $h/temp3_input 2>/dev/null)"
    done
    echo "$line"; sleep 0.2
  done ) > $O/clocks_$ID.txt 2>/dev/null &
SP=$!
sleep 0.5
timeout 120 ./scripts/kbench_loop 512 $S > $O/loop_$ID.txt 2>&1
kill $SP 2>/dev/null
python3 - $O/clocks_$ID.txt $O/loop_$ID.txt "$TAG" <<'PY'
import sys
rows = [l.split('|') for l in open(sys.argv[1]) if '|' in l]
ms = sorted(float(l.split()[1]) for l in open(sys.argv[2]) if l[0].isdigit())
cp = sorted(float(l.split()[2]) for l in open(sys.argv[2]) if l[0].isdigit())
va = sorted(float(l.split()[4]) for l in open(sys.argv[2]) if l[0].isdigit())
nd = min(len(r) for r in rows) - 1
peak = [max(float(r[1 + i].split()[1]) for r in rows) for i in range(nd)]
ours = max(range(nd), key=lambda i: peak[i])
busy = [r[1 + ours].split() for r in rows if float(r[1 + ours].split()[1]) > 0.6 * peak[ours]]
med = lambda v: sorted(v)[len(v) // 2]
others = sum(1 for i in range(nd) if i != ours and peak[i] > 600e6)
print(f"[clock probe {sys.argv[3]}] k_fused3d<64,8,8> median {med(ms):.3f} ms (min {ms[0]:.3f}, max {ms[-1]:.3f}); copy {med(cp):.3f} ms; fp64 VALU {med(va):.2f} ms | sclk {med([float(b[0]) for b in busy]) / 1e6:.0f} MHz, "
      f"power {med([float(b[1]) for b in busy]) / 1e6:.0f} W, junction {med([float(b[2]) for b in busy]) / 1e3:.0f} C, HBM {med([float(b[3]) for b in busy]) / 1e3:.0f} C | {nd} devices visible, {others} others above 600 W")
PY
