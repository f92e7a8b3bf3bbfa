#!/usr/bin/env python3
"""Tiles per wave x grid cap of k_decode_flat_lut at 2^16 .. 2^20 frames and the
headline's 1069463 (fresh tensors per size; every setting on the same tensors)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
FN, PN, SPF = 8032, 8000, 32000
CODER = _lib.CODER_VDIF
settings = [('default', 12, 0)] + [('tpw%d cap2^%d' % (t, c), t, 1 << c) for t in (4, 5, 6, 8) for c in (21, 22, 23, 25)] \
    + [('default again', 12, 0)]
if len(sys.argv) > 1:
    FN, PN, SPF = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[2]) * 4
for nfr in (int(8 * 2 ** 30) // FN, (1 << 28) * 8 // FN):
    buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
    out = torch.empty(nfr * SPF, dtype=torch.float32, device=dev)
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
    res = {}
    for name, tpw, cap in settings:
        kernels.tune(_lib.TUNE_TILES_PER_WAVE, tpw)
        kernels.tune(_lib.TUNE_BLOCKS, cap)
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, _lib.CODER_VDIF, 2, src=src, out=out), reps=5)
        res[name] = round(nfr * (FN + SPF * 4) / ms / 1e9, 3)
    kernels.tune(_lib.TUNE_TILES_PER_WAVE, 12)
    kernels.tune(_lib.TUNE_BLOCKS, 0)
    print(json.dumps(dict(frames=nfr, TBps=res)), flush=True)
    del buf, out, src
    torch.cuda.empty_cache()
