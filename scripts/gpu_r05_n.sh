#!/bin/bash
# round 5, call N: the kernel's rate beside the device's clocks / power, second by second
mkdir -p gpurun_out/r05n
O=gpurun_out/r05n
ls /sys/class/drm/ > $O/sysfs.txt 2>&1
for d in /sys/class/drm/card*/device; do echo "== $d" >> $O/sysfs.txt; ls $d | tr '\n' ' ' >> $O/sysfs.txt; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' ' >> $O/sysfs.txt; echo >> $O/sysfs.txt; done
rocm-smi --showclocks --showpower --showtemp --showperflevel --showmaxpower 2>&1 | grep -v "^=\|^$" | head -40 > $O/smi_idle.txt
sampler() {
  while true; do
    t=$(date +%s.%N | cut -c1-13)
    line="$t"
    for d in /sys/class/drm/card*/device; do
      for f in $d/hwmon/hwmon*/freq1_input $d/hwmon/hwmon*/freq2_input $d/hwmon/hwmon*/power1_average $d/hwmon/hwmon*/power1_input $d/hwmon/hwmon*/temp1_input $d/hwmon/hwmon*/temp2_input $d/hwmon/hwmon*/temp3_input $d/hwmon/hwmon*/power1_cap; do
        [ -r $f ] && line="$line $(basename $f)=$(cat $f 2>/dev/null)"
      done
      [ -r $d/pp_dpm_sclk ] && line="$line sclk=[$(grep '\*' $d/pp_dpm_sclk 2>/dev/null | tr -d '\n')]"
      [ -r $d/pp_dpm_mclk ] && line="$line mclk=[$(grep '\*' $d/pp_dpm_mclk 2>/dev/null | tr -d '\n')]"
      [ -r $d/pp_dpm_fclk ] && line="$line fclk=[$(grep '\*' $d/pp_dpm_fclk 2>/dev/null | tr -d '\n')]"
      [ -r $d/gpu_busy_percent ] && line="$line busy=$(cat $d/gpu_busy_percent 2>/dev/null)"
    done
    echo "$line"
    sleep 0.2
  done
}
sampler > $O/clocks.txt 2>&1 &
SP=$!
sleep 1
for i in 1 2 3; do timeout 120 ./scripts/kbench_loop 512 12 > $O/loop_$i.txt 2>&1; sleep 2; done
( for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' ' | sed 's/GPU\[[0-9]\]\s*: //g; s/  */ /g'; echo; sleep 1; done ) > $O/smi_busy.txt 2>&1 &
SM=$!
timeout 120 ./scripts/kbench_loop 512 10 > $O/loop_4.txt 2>&1
wait $SM
kill $SP
head -3 $O/clocks.txt | cut -c1-400; wc -l $O/clocks.txt
for i in 1 2 3 4; do awk 'NR>1 {print $2}' $O/loop_$i.txt | sort -n | awk '{a[NR]=$1} END {print "loop: min", a[1], "median", a[int((NR+1)/2)], "max", a[NR], "n", NR}'; done
cat $O/smi_busy.txt | cut -c1-300
head -12 $O/sysfs.txt | cut -c1-600
