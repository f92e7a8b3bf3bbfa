#!/usr/bin/env python3
"""Launch-size dependence of the other persistent kernels (thread interleave
rows / gather, Mark 4): default grid cap against an uncapped grid."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nbytes * 4, dtype=torch.float32, device='cuda')
fn_, pn, nth = 8032, 8000, 8
m4 = BITMAPS[(8, 2, 4)]
for lg in range(13, 18):
    nsets = 1 << lg                       # frame sets of 8 threads
    src = (torch.arange(nsets * nth, device='cuda', dtype=torch.int64) * fn_ + 32)
    alg = nsets * nth * (fn_ + pn * 16)
    row = dict(frames=nsets * nth)
    for name, chunk in (('rows_chunk32', 32), ('gather_chunk4', 4), ('gather_chunk1', 1)):
        for blocks, tag in ((0, 'cap'), (1 << 30, 'uncapped')):
            kernels.tune(_lib.TUNE_BLOCKS, blocks)
            ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=chunk, nslot=nth, src=src,
                                                      out=out[:nsets * nth * pn * 4]), reps=5)
            row[name + '_' + tag] = round(alg / ms / 1e9, 2)
    nfr = nsets * nth * fn_ // 160000
    nout = nfr * 20000 * 32
    for blocks, tag in ((0, 'cap'), (1 << 30, 'uncapped')):
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(lambda: kernels.decode_mark4(buf, nfr, 64, 20000, m4['sign_bit'], m4['mag_bit'], fill_words=160,
                                                 src0=0, src_stride=160000, out=out[:nout]), reps=5)
        row['mark4_' + tag] = round((nfr * 160000 + nout * 4) / ms / 1e9, 2)
    print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_BLOCKS, 0)
