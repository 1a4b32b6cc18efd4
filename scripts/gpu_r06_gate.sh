#!/bin/bash
# round 6 gate run: the full GPU suite in the driver's order (-x as the driver runs it), smoke, then the driver's bench command
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r06_gate.sh TAG'
tag=${1:-r06a}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
for i in 1 2; do
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench$i.json 2> $out/bench$i.err
  cp bench_details.json $out/bench${i}_details.json 2>/dev/null
  wc -c $out/bench$i.json; cat $out/bench$i.json; tail -4 $out/bench$i.err
done
