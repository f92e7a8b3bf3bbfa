#!/usr/bin/env python3
"""Workgroups per CU of k_decode_flat_lut (bounded by unused LDS, BB_TUNE_LDS_PAD)
at 8 and 2 GiB of cfg2 frames, same tensors: do fewer concurrent write streams help?"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
FN, PN = 8032, 8000
for gib in (8, 2):
    nfr = (gib << 30) // FN
    buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
    out = torch.empty(nfr * PN * 4, dtype=torch.float32, device=dev)
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
    for tiles in (4, 8):
        kernels.tune(_lib.TUNE_LUT_TILES, tiles)
        res = {}
        for pad in (0, 4096, 8192, 12288, 16384, 24576, 36864, 49152, 65536, 0):
            kernels.tune(_lib.TUNE_LDS_PAD, pad)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, 0, 2, src=src, out=out), reps=8)
            res['pad %d (<= %d WG/CU)%s' % (pad, min(16, (160 << 10) // (4096 + pad)), ' again' if pad == 0 and res else '')] = \
                round(nfr * (FN + PN * 16) / ms / 1e9, 3)
        kernels.tune(_lib.TUNE_LDS_PAD, 0)
        print(json.dumps(dict(GiB=gib, tiles_per_wave=tiles, TBps=res)), flush=True)
    kernels.tune(_lib.TUNE_LUT_TILES, 4)
    del buf, out, src
    torch.cuda.empty_cache()
