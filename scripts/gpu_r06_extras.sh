#!/bin/bash
# round 6: the legs behind the contract line (bench.py --extras -> bench_details.json) and the N > 1 control flow on one device (two ranks on device 0, ipc transport)
out=gpurun_out/${1:-r06g}; mkdir -p $out
( time python bench.py --gpus 1 --steps 20 --warmup 5 --extras --details $out/extras_details.json ) > $out/extras.json 2> $out/extras.err
wc -c $out/extras.json; tail -3 $out/extras.err | cut -c1-300
( time python bench.py --gpus 2 --same-device --default-transport ipc --n 256 --steps 20 --warmup 5 --details $out/two_details.json ) > $out/two.json 2> $out/two.err
cat $out/two.json | cut -c1-1500; tail -3 $out/two.err | cut -c1-300
( time python bench.py --gpus 2 --same-device --default-transport ipc --n 256 --steps 20 --warmup 5 --extras --extras-budget 300 --details $out/two_extras_details.json ) > $out/two_extras.json 2> $out/two_extras.err
wc -c $out/two_extras.json; tail -3 $out/two_extras.err | cut -c1-300
