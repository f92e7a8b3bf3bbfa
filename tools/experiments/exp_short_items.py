#!/usr/bin/env python3
"""Short work items on an uncapped grid for the other persistent kernels
(k_decode_rows_pipe: cfg3 layout; k_decode_gather: 8 threads x 1 channel;
k_decode_mark4), 8 GiB and 2 GiB inputs, every setting on the same tensors."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
m = BITMAPS[(8, 2, 4)]
settings = [('default', 12, 0), ('cap 2^23', 12, 1 << 23), ('tpw4 cap 2^23', 4, 1 << 23), ('tpw2 cap 2^23', 2, 1 << 23),
            ('tpw4', 4, 0), ('default again', 12, 0)]
for gib in (8, 2):
    nbytes = gib << 30
    buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device=dev)
    out = torch.empty(nbytes * 4, dtype=torch.float32, device=dev)
    nsets = nbytes // 8032 // 8
    src8 = (torch.arange(nsets * 8, device=dev, dtype=torch.int64) * 8032 + 32)
    nf4 = nbytes // 160000
    cases = {
        'rows (8 thr x 16 ch complex)': (lambda: kernels.decode_frames(buf, nsets, 8000, 0, 2, chunk=32, nslot=8, src=src8,
                                                                         complex_data=True, out=out), nsets * 8 * (8032 + 128000)),
        'gather (8 thr x 1 ch)': (lambda: kernels.decode_frames(buf, nsets, 8000, 0, 2, chunk=1, nslot=8, src=src8, out=out),
                                  nsets * 8 * (8032 + 128000)),
        'mark4 64 tracks': (lambda: kernels.decode_mark4(buf, nf4, 64, 20000, m['sign_bit'], m['mag_bit'], fill_words=160,
                                                         src0=0, src_stride=160000, out=out), nf4 * (160000 + 2560000)),
    }
    for cname, (fn, nb) in cases.items():
        res = {}
        for name, tpw, cap in settings:
            kernels.tune(_lib.TUNE_TILES_PER_WAVE, tpw)
            kernels.tune(_lib.TUNE_BLOCKS, cap)
            res[name] = round(nb / timeit(fn, reps=5) / 1e9, 3)
        kernels.tune(_lib.TUNE_TILES_PER_WAVE, 12)
        kernels.tune(_lib.TUNE_BLOCKS, 0)
        print(json.dumps(dict(GiB=gib, case=cname, kernel=_lib.last_kernel()[:60], TBps=res)), flush=True)
    del buf, out, src8
    torch.cuda.empty_cache()
