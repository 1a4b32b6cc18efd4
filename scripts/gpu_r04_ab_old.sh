#!/bin/bash
# A/B on ONE box: the tree before the body-force forms (_old/, commit b24392c, built beside) against the current tree, headline leg alone, alternating.
# _old/ is not kept in the repository:  mkdir _old && git archive b24392c | tar -x -C _old && (cd _old && python justrelax.jl_amd/build.py && python __graft_entry__.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04ab}
mkdir -p $OUT
for r in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/_old && timeout 600 python bench.py --no-extras --no-cpu-baseline --no-steady-state > $OUT/old_$r.json 2> $OUT/old_$r.err)
  (cd $GRAFT_REPO_ROOT && timeout 600 python bench.py --no-extras --no-cpu-baseline --no-steady-state > $OUT/new_$r.json 2> $OUT/new_$r.err)
done
python - <<PY
import json
for r in (1, 2, 3):
    for t in ("old", "new"):
        try:
            d = json.loads(open("$OUT/%s_%d.json" % (t, r)).read().strip().splitlines()[-1])
        except Exception as e:
            print(t, r, "failed", e); continue
        g = d.get("general_kernel", {}); b = d.get("with_body_forces", {})
        print(t, r, "headline %.1f it/s kernel %.3f ms" % (d["value"], d["roofline"]["avg_launch_ms"]), d["config"]["kernel_form"],
              "| with forces %s" % (b.get("roofline") or {}).get("avg_launch_ms"), "| general %s" % (g.get("roofline") or {}).get("avg_launch_ms"))
PY
