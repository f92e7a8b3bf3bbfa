#!/usr/bin/env python3
"""Mark 4 decode and encode for every supported mode (ntrack 16 / 32 / 64)."""
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
for key, m in BITMAPS.items():
    ntrack = m['ntrack']
    wbytes = ntrack // 8
    frame = 20000 * wbytes
    nfr = nbytes // frame
    nwords = 20000
    nout = nfr * nwords * (ntrack // 2)
    alg = nfr * frame + nout * 4
    ms = timeit(lambda: kernels.decode_mark4(buf, nfr, ntrack, nwords, m['sign_bit'], m['mag_bit'], fill_words=160,
                                             src0=0, src_stride=frame, out=out[:nout]), reps=5)
    row = dict(mode=str(key if key[1] < 100 else (key[0], 'Ft', key[2])), ntrack=ntrack,
               decode_TBps=round(alg / ms / 1e9, 2))
    vals = out[:nout]
    ms = timeit(lambda: kernels.encode_mark4(vals, ntrack, m['sign_bit'], m['mag_bit']), reps=5)
    row['encode_TBps'] = round((nout * 4 + nout // (ntrack // 2) * wbytes) / ms / 1e9, 2)
    print(json.dumps(row), flush=True)
