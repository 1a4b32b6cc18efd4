"""two blocks on one device through the in-process transport: the bench leg `multi_rank_path` alone (python scripts/bench_multi_rank.py [n] [steps] [x|y|z|xyz serial|overlap] | vep x|z)"""
import json
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
import bench
jr = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
only = tuple(sys.argv[3:5]) if len(sys.argv) > 4 else None
print(json.dumps(bench.cfg_multi_rank_path(jr, n=n, steps=steps, only=only), indent=1))
