#!/usr/bin/env python3
"""Headline launch (2^20-frame cfg2 decode) with the output tensor allocated
before / after the input image.  One setting per process:
    python tools/experiments/exp_alloc_order.py in-first|out-first"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
order = sys.argv[1]
kernels.init()
dev = torch.device('cuda')
FN, PN, SPF = 8032, 8000, 32000
nfr = int(8 * 2 ** 30) // FN
if order == 'out-first':
    out = torch.empty(nfr * SPF, dtype=torch.float32, device=dev)
    buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
else:
    buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
    out = torch.empty(nfr * SPF, dtype=torch.float32, device=dev)
src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
ms = [timeit(lambda: kernels.decode_frames(buf, nfr, PN, _lib.CODER_VDIF, 2, src=src, out=out), reps=10) for _ in range(2)]
print(json.dumps(dict(order=order, TBps=[round(nfr * (FN + SPF * 4) / m / 1e9, 3) for m in ms],
                      out_ptr=hex(out.data_ptr()), in_ptr=hex(buf.data_ptr()))), flush=True)
