#!/usr/bin/env python3
"""Plain flat kernel for 8-bit samples (k_decode_flat<8,...>): tiles per workgroup."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
for gib in (31, 8):
    nb = gib << 30
    buf = torch.randint(0, 256, (nb + 8192,), dtype=torch.uint8, device=dev)
    out = torch.empty(nb, dtype=torch.float32, device=dev)
    for coder, name in ((_lib.CODER_INT, 'int8'), (_lib.CODER_VDIF, 'vdif 8-bit')):
        res = {}
        for seg in (32, 8, 16, 64, 128, 32):
            kernels.tune(_lib.TUNE_SEG_TILES, seg)
            ms = timeit(lambda: kernels.decode_frames(buf, 1, nb, coder, 8, src0=0, out=out), reps=4)
            res['%d tiles%s' % (seg, ' again' if '%d tiles' % seg in res else '')] = round(nb * 5 / ms / 1e9, 3)
        kernels.tune(_lib.TUNE_SEG_TILES, 0)
        print(json.dumps(dict(GiB=gib, coder=name, kernel=_lib.last_kernel()[:50], TBps=res)), flush=True)
    del buf, out
    torch.cuda.empty_cache()
