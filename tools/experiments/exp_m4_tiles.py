#!/usr/bin/env python3
"""Mark 4 decode: tiles per wave and work item (4 waves per workgroup)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
for key, nt in (((8, 2, 4), 64), ((2, 2, 4), 16)):
    m = BITMAPS[key]
    fb = nt * 2500
    for gib in (8, 2):
        nbytes = gib << 30
        nf = nbytes // fb
        buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
        out = torch.empty(nf * fb * 4, dtype=torch.float32, device=dev)
        res = {}
        for tiles in (8, 4, 2, 1, 6, 8):
            kernels.tune(_lib.TUNE_M4_TILES, tiles)
            ms = timeit(lambda: kernels.decode_mark4(buf, nf, nt, 20000, m['sign_bit'], m['mag_bit'], fill_words=160,
                                                     src0=0, src_stride=fb, out=out), reps=6)
            res['%d tiles%s' % (tiles, ' again' if '%d tiles' % tiles in res else '')] = round(nf * fb * 17 / ms / 1e9, 3)
        kernels.tune(_lib.TUNE_M4_TILES, 8)
        print(json.dumps(dict(ntrack=nt, GiB=gib, kernel=_lib.last_kernel()[:70], TBps=res)), flush=True)
        del buf, out
        torch.cuda.empty_cache()
