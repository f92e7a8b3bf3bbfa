#!/usr/bin/env python3
"""k_decode_i8_xpose: 128 against 64 output rows per tile at the headline's output size."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
for gib in (31, 8):
    nbytes = gib << 30
    buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device=dev)
    out = torch.empty(nbytes, dtype=torch.float32, device=dev)
    npol, nchan, blk = 2, 64, 128 << 20
    T = blk // (npol * nchan * 2)
    nf = nbytes // blk
    for layout, name in ((_lib.LAYOUT_GUPPI_CF, 'CF'), (_lib.LAYOUT_GUPPI_TF, 'TF'), (_lib.LAYOUT_MKBF, 'MKBF')):
        res = {}
        for rows in (128, 64, 128, 64):
            kernels.tune(_lib.TUNE_XPOSE_ROWS, rows)
            ms = timeit(lambda: kernels.decode_i8_tiled(buf, nf, layout, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=out), reps=4)
            res['%d rows%s' % (rows, ' again' if '%d rows' % rows in res else '')] = round(nf * blk * 5 / ms / 1e9, 3)
        kernels.tune(_lib.TUNE_XPOSE_ROWS, 0)
        print(json.dumps(dict(GiB=gib, layout=name, TBps=res)), flush=True)
    del buf, out
    torch.cuda.empty_cache()
