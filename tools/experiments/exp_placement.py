#!/usr/bin/env python3
"""baseband_amd.empty_output against torch.empty for outputs of 2^16 .. 2^19 cfg2
frames: decode rate into the tensor each of them returns (fresh process state per
size: everything is freed in between)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import baseband_amd
from baseband_amd import kernels
from baseband_amd.placement import probe_rate
kernels.init()
for rep in range(2):
    for lg in (16, 17, 18, 19):
        n = (1 << lg) * 32000
        plain = torch.empty(n, dtype=torch.float32, device='cuda')
        r_plain = probe_rate(plain)
        del plain
        torch.cuda.empty_cache()
        rates = []
        best = baseband_amd.empty_output((n,), candidates=4, report=rates)
        r_best = probe_rate(best)
        del best
        torch.cuda.empty_cache()
        print(json.dumps(dict(rep=rep, frames=1 << lg, out_GB=round(n * 4 / 1e9, 1), torch_empty_TBps=round(r_plain, 3),
                              candidates_TBps=[round(r, 3) for r in rates], empty_output_TBps=round(r_best, 3))), flush=True)
