"""The calibrated gather probes of bench.py (gather_probe_16m: 16 M lookups per launch, sequential-id calibration first),
stand-alone: python tools/gather_probe16m.py [out.json]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from hpmn_amd import build
build.build_library()
res = bench.gather_probe_16m(torch.device("cuda:0"), reps=12)
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
