#!/usr/bin/env python3
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
from tools.bench_formats import timeit
kernels.init()
n = (32 << 30) // 4
x = torch.randn(n, dtype=torch.float32, device='cuda')
m = BITMAPS[(8, 2, 4)]
for name, fn in (('flat 2-bit', lambda: kernels.encode_flat(x, 0, 2)), ('flat 8-bit', lambda: kernels.encode_flat(x, 0, 8)),
                 ('mark4', lambda: kernels.encode_mark4(x, 64, m['sign_bit'], m['mag_bit']))):
    for blocks in (262144, 524288, 1048576, 2097152, 4194304, 8388608, 16777216):
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(fn)
        print(json.dumps(dict(case=name, blocks=blocks, ms=round(ms, 3), read_TBps=round(n * 4 / ms / 1e9, 3))), flush=True)
