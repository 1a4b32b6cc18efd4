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
# static / slow-changing state of every device: partition modes, DPM levels (read while idle), link / memory info
( for d in /sys/class/drm/card*/device; do
    [ -r $d/current_compute_partition ] || continue
    echo "$(basename $(dirname $d)): compute_partition=$(cat $d/current_compute_partition 2>/dev/null) memory_partition=$(cat $d/current_memory_partition 2>/dev/null) fclk=[$(tr '\n' ' ' < $d/pp_dpm_fclk 2>/dev/null)] mclk=[$(tr '\n' ' ' < $d/pp_dpm_mclk 2>/dev/null)] socclk=[$(tr '\n' ' ' < $d/pp_dpm_socclk 2>/dev/null)] perf=$(cat $d/power_dpm_force_performance_level 2>/dev/null) vram_total=$(cat $d/mem_info_vram_total 2>/dev/null) vram_used=$(cat $d/mem_info_vram_used 2>/dev/null) xgmi_hive=$(cat $d/xgmi_hive_info/xgmi_hive_id 2>/dev/null) numa=$(cat $d/numa_node 2>/dev/null)"
  done ) > $O/static_$ID.txt 2>&1
for d in /sys/class/drm/card*/device; do [ -r $d/unique_id ] && echo "$(basename $(dirname $d)) unique_id=$(cat $d/unique_id) bus=$(basename $(realpath $d)) vram_used=$(cat $d/mem_info_vram_used 2>/dev/null)"; done > $O/ids_$ID.txt 2>&1
echo "host $(hostname) ; devices: $(awk '{print $2, $4}' $O/ids_$ID.txt | tr '\n' ';' | cut -c1-600)"
grep -E "^# device|distinct" $O/loop_$ID.txt | cut -c1-300
python3 - $O/clocks_$ID.txt $O/loop_$ID.txt "$TAG" <<'PY'
import sys
rows = [l.split('|') for l in open(sys.argv[1]) if '|' in l]
ms = sorted(float(l.split()[1]) for l in open(sys.argv[2]) if l[0].isdigit())
cp = sorted(float(l.split()[2]) for l in open(sys.argv[2]) if l[0].isdigit())
va = sorted(float(l.split()[4]) for l in open(sys.argv[2]) if l[0].isdigit())
mall = sorted(float(l.split()[5]) for l in open(sys.argv[2]) if l[0].isdigit())
hbm = sorted(float(l.split()[6]) for l in open(sys.argv[2]) if l[0].isdigit())
lat1 = sorted(float(l.split()[7]) for l in open(sys.argv[2]) if l[0].isdigit())
latm = sorted(float(l.split()[8]) for l in open(sys.argv[2]) if l[0].isdigit())
nd = min(len(r) for r in rows) - 1
peak = [max(float(r[1 + i].split()[1]) for r in rows) for i in range(nd)]
ours = max(range(nd), key=lambda i: peak[i])
busy = [r[1 + ours].split() for r in rows if float(r[1 + ours].split()[1]) > 0.6 * peak[ours]]
med = lambda v: sorted(v)[len(v) // 2]
others = sum(1 for i in range(nd) if i != ours and peak[i] > 600e6)
print(f"[clock probe {sys.argv[3]}] k_fused3d<64,8,8> median {med(ms):.3f} ms (min {ms[0]:.3f}, max {ms[-1]:.3f}); copy {med(cp):.3f} ms; fp64 VALU {med(va):.2f} ms; re-read of 128 MiB {med(mall):.0f} GB/s; read of 1 GiB {med(hbm):.0f} GB/s; dependent load {med(lat1):.0f} ns (one lane), {med(latm):.0f} ns (2,048 lanes) | sclk {med([float(b[0]) for b in busy]) / 1e6:.0f} MHz, "
      f"power {med([float(b[1]) for b in busy]) / 1e6:.0f} W, junction {med([float(b[2]) for b in busy]) / 1e3:.0f} C, HBM {med([float(b[3]) for b in busy]) / 1e3:.0f} C | {nd} devices visible, {others} others above 600 W")
PY
