#!/usr/bin/env python3
"""Yardstick: pure HBM read rate (torch reductions) next to the 2-bit encoder."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
n = (32 << 30) // 4
x = torch.randn(n, dtype=torch.float32, device='cuda')
for name, fn in (('torch.sum', lambda: x.sum()), ('torch.max', lambda: x.max()),
                 ('torch.count_nonzero(x > 0)', lambda: torch.count_nonzero(x > 0)),
                 ('k_encode_flat 2-bit', lambda: kernels.encode_flat(x, 0, 2)),
                 ('k_encode_flat 1-bit', lambda: kernels.encode_flat(x, 0, 1))):
    ms = timeit(fn)
    print(json.dumps(dict(case=name, ms=round(ms, 3), read_TBps=round(n * 4 / ms / 1e9, 3))), flush=True)
for blocks in (1024, 2048, 4096, 8192, 16384):
    kernels.tune(_lib.TUNE_BLOCKS, blocks)
    ms = timeit(lambda: kernels.encode_flat(x, 0, 2))
    print(json.dumps(dict(case='k_encode_flat 2-bit, TUNE_BLOCKS=%d' % blocks, ms=round(ms, 3),
                          read_TBps=round(n * 4 / ms / 1e9, 3))), flush=True)
