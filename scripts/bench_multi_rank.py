"""two blocks on one device through the in-process transport: the bench leg `multi_rank_path` alone
   python scripts/bench_multi_rank.py [n] [steps] [x|y|z|xyz default|serial|early|overlap] [KEY=INT ...]   |   ... vep x|z"""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
import bench_extras as bench
jr = load_package()
args = [a for a in sys.argv[1:] if "=" not in a]
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
n = int(args[0]) if len(args) > 0 else 512
steps = int(args[1]) if len(args) > 1 else 40
only = tuple(args[2:4]) if len(args) > 3 else None
print(json.dumps(bench.cfg_multi_rank_path(jr, n=n, steps=steps, only=only, handle_options=opts or None), indent=1))
