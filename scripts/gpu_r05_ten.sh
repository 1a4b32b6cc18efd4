#!/bin/bash
# VERDICT r4 item 1, "done" criterion: ten fresh processes, k_fused3d launch time at 512^3 of each (the driver's command without the extra legs); beside it what the same
# process's allocation would have run at without the search (ms per iteration of a 12-iteration batch as allocated / after the search) and one process with the search switched off
T=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05ten_$T; mkdir -p $OUT
python3 -c "import torch; p=torch.cuda.get_device_properties(0); print('device', p.name, 'pci %02x:%02x' % (p.pci_bus_id, p.pci_device_id))"
for i in 1 2 3 4 5 6 7 8 9 10; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-general-kernel > $OUT/p$i.json 2> /dev/null
done
for i in 11 12; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-general-kernel --placement-draws 0 > $OUT/p$i.json 2> /dev/null
done
python3 - <<PY
import json
for i in range(1, 13):
    d = json.load(open("$OUT/p%d.json" % i)); r = d["roofline"]; ps = r.get("placement_search") or {}
    print("process %2d: %6.1f it/s (20 steps) %6.1f (100 steps)  k_fused3d %.3f ms  | %s" % (i, d["value"], (d.get("steady_state") or {}).get("value") or 0.0, r["avg_launch_ms"],
          ("search: %.3f ms as allocated -> %.3f, %d of %d draws kept, %.1f s" % (ps["ms_as_allocated"], ps["ms_kept"], ps["kept"], ps["draws"], ps["seconds"])) if ps.get("draws") else "no search (--placement-draws 0, torch's arrays)"))
PY
