#!/usr/bin/env python3
"""4-bit samples: byte table kernel (k_decode_flat_lut<4>) against the plain
kernel (BB_TUNE_BYTE_LUT 0), VDIF levels and GSB nibbles, same tensors."""
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
    out = torch.empty(nfr * PN * 2, dtype=torch.float32, device=dev)
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
    for coder in (_lib.CODER_VDIF, _lib.CODER_INT):
        res = {}
        for name, lut, tiles in (('plain', 0, 4), ('byte table 4 tiles', 1, 4), ('byte table 8 tiles', 1, 8),
                                 ('byte table 12 tiles', 1, 12), ('byte table 16 tiles', 1, 16), ('plain again', 0, 4)):
            kernels.tune(_lib.TUNE_BYTE_LUT, lut)
            kernels.tune(_lib.TUNE_LUT_TILES, tiles)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, coder, 4, src=src, out=out), reps=6)
            res[name] = round(nfr * (FN + PN * 8) / ms / 1e9, 3)
        kernels.tune(_lib.TUNE_BYTE_LUT, 1)
        kernels.tune(_lib.TUNE_LUT_TILES, 4)
        print(json.dumps(dict(GiB=gib, coder=coder, kernel=_lib.last_kernel()[:60], TBps=res)), flush=True)
    del buf, out, src
    torch.cuda.empty_cache()
