#!/bin/bash
mkdir -p gpurun_out/r05q; O=gpurun_out/r05q
which amd-smi rocm-smi > $O/tools.txt 2>&1
amd-smi metric -h > $O/amdsmi_metric_help.txt 2>&1
rocm-smi --help 2>&1 | grep -i -E "metric|violation|throttl|voltage|power" > $O/rocmsmi_help.txt
( ./scripts/kbench_loop 512 8 > $O/loop.txt 2>&1 ) &
sleep 4
amd-smi metric --json > $O/amdsmi_metric_busy.json 2>&1
rocm-smi --showmetrics > $O/rocmsmi_metrics_busy.txt 2>&1
wait
grep -E "^# device" $O/loop.txt | cut -c1-200
awk 'NR>6 {print $2}' $O/loop.txt | sort -n | awk '{a[NR]=$1} END {print "fused median ms", a[int((NR+1)/2)]}'
head -c 1500 $O/amdsmi_metric_help.txt
echo; wc -c $O/amdsmi_metric_busy.json $O/rocmsmi_metrics_busy.txt
