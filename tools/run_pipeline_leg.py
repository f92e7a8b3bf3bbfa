"""bench.py's `pipeline` leg alone:  python tools/run_pipeline_leg.py [GiB per file]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from baseband_amd import kernels
kernels.init()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
print(json.dumps(bench.leg_pipeline(torch.device('cuda', 0), gib=gib)))
