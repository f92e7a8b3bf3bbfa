#!/usr/bin/env python3
"""1-bit samples: tiles per wave of the byte table kernel (same tensors)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
FN, PN = 8032, 8000
for gib in (4, 1):
    nfr = (gib << 30) // FN
    buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
    out = torch.empty(nfr * PN * 8, dtype=torch.float32, device=dev)
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
    res = {}
    for tiles in (1, 2, 3, 4, 6, 8, 4):
        kernels.tune(_lib.TUNE_LUT_TILES, tiles)
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, 0, 1, src=src, out=out), reps=6)
        res['%d tiles%s' % (tiles, ' again' if '%d tiles' % tiles in res else '')] = round(nfr * (FN + PN * 32) / ms / 1e9, 3)
    kernels.tune(_lib.TUNE_LUT_TILES, 4)
    print(json.dumps(dict(GiB=gib, kernel=_lib.last_kernel()[:60], TBps=res)), flush=True)
    del buf, out, src
    torch.cuda.empty_cache()
