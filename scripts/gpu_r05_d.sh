#!/bin/bash
# round 5, call D: 64 x 8 tile + operand cache + field pool in the library: parity tests, then the default bench line and an A/B of the tile shapes
mkdir -p gpurun_out/r05d
python -m pytest tests/test_gpu_field_alloc.py tests/test_gpu_stokes3d.py tests/test_gpu_two_blocks.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r05d/tests.log 2>&1
tail -3 gpurun_out/r05d/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05d/bench_default.json 2> gpurun_out/r05d/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05d/bench_default.json'))
r=d['roofline']
print('value', d['value'], 'steady', d['steady_state']['value'] if d.get('steady_state') else None, 'kernel ms', r['avg_launch_ms'], 'frac', r['frac'], 'needed frac', r.get('frac_at_needed_bytes'))
print('general', r.get('general_form'))
oc=d.get('other_configs',{})
for k,v in oc.items():
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if isinstance(vv,(int,float,str)) and kk!='workload'})
mr=oc.get('multi_rank_path',{})
for k in ('split_x','split_z','vep3d_256_split_z'):
    print(k, json.dumps(mr.get(k))[:900])
ipc=mr.get('ipc_two_processes',{})
for k in ('split_x','split_z'):
    v=ipc.get(k,{})
    print('ipc',k,{m:{kk:vv for kk,vv in (v.get(m) or {}).items() if kk!='chain_us_per_rank'} for m in ('default','early','serial')})
PY
for i in 1 2 3; do python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline --option fused_tile=0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile 64x4: value %.1f kernel %.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))"
python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile auto (64x8): value %.1f kernel %.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))"; done
