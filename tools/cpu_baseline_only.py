"""The bench's cpu_baseline leg alone (reproducibility check): python tools/cpu_baseline_only.py [config]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
c = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"])
r = bench.cpu_baseline(c)
print(json.dumps({k: r[k] for k in r if k in ("value", "cores", "train_step_seconds", "malloc_tuned", "sample")}))
