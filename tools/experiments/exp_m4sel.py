#!/usr/bin/env python3
"""Mark 4 full decode: the pipelined persistent kernel (k_decode_mark4) against
the LDS-staged one-item-per-workgroup kernel written for channel selections
(k_decode_mark4_select with the complete maps), same process, 0.5 .. 8 GiB of
64-track fanout-4 frames in HBM."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
out = torch.empty(nbytes // 160000 * 640000, dtype=torch.float32, device=dev)
for key in ((8, 2, 4), (4, 2, 4), (2, 2, 4)):
    maps = BITMAPS[key]
    nt = {8: 64, 4: 32, 2: 16}[key[0]]
    fb = nt * 2500
    per = nt // 2 * 20000
    for gib in (0.5, 2, 4, 8):
        nf = int(gib * 2 ** 30) // fb
        res = {}
        for name, sel in (('pipelined', False), ('lds', True), ('pipelined2', False), ('lds2', True)):
            ms = timeit(lambda: kernels.decode_mark4(buf, nf, nt, 20000, maps['sign_bit'], maps['mag_bit'],
                                                     fill_words=160, src0=0, src_stride=fb,
                                                     out=out[:nf * per], select=sel), reps=5)
            res[name] = round(nf * (fb + per * 4) / ms / 1e9, 3)
        print(json.dumps(dict(ntrack=nt, GiB_in=gib, frames=nf, TBps=res)), flush=True)
