#!/bin/bash
# several processes in a row, each: kernel time of the headline form + plain streaming probes; rocm-smi clocks / power sampled beside every process
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04two}
mkdir -p $OUT
for r in 1 2 3 4 5 6 7 8 9 10; do
  ( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 1; done ) > $OUT/smi_$r.txt 2>/dev/null &
  SMI=$!
  timeout 300 python3 scripts/probe_two_rates.py > $OUT/p_$r.txt 2> $OUT/p_$r.err
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  echo "process $r: $(cat $OUT/p_$r.txt)"
  echo "   smi (mid-run sample): $(sed -n '12p' $OUT/smi_$r.txt | cut -c1-300)"
done
