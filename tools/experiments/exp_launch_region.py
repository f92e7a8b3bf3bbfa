#!/usr/bin/env python3
"""Slow vs fast output regions of one big buffer: which ingredient of the decode
kernel is sensitive to where the output lies?"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
payload, header = 8000, 32
stride = payload + header
nmax = 1 << 20
buf = torch.randint(0, 256, (nmax * stride + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nmax * payload * 4, dtype=torch.float32, device='cuda')
per = payload * 4
nfr = 1 << 17
alg = nfr * (stride + payload * 16)
regions = [k * nfr for k in range(8)]

def dec(f0):
    o = out[f0 * per:(f0 + nfr) * per]
    return timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2,
                                                src0=header, src_stride=stride, out=o), reps=5)

def row(tag):
    print(json.dumps({"tag": tag, "TBps_by_region": [round(alg / dec(f0) / 1e9, 2) for f0 in regions]}), flush=True)

row('default')
kernels.tune(_lib.TUNE_NT_STORES, 0); row('plain stores'); kernels.tune(_lib.TUNE_NT_STORES, 1)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 0); row('variant 0 (plain kernel)')
kernels.tune(_lib.TUNE_NT_STORES, 0); row('variant 0, plain stores'); kernels.tune(_lib.TUNE_NT_STORES, 1)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 3); row('variant 3 (pipelined, unaligned loads)')
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
kernels.tune(_lib.TUNE_BLOCKS, 8192); row('grid 8192'); kernels.tune(_lib.TUNE_BLOCKS, 0)
fills, copies = [], []
src = out[(nmax - nfr // 4) * per:]                       # 4.2 GB source for the copy
for f0 in regions:
    o = out[f0 * per:(f0 + nfr) * per]
    fills.append(round(o.numel() * 4 / timeit(lambda: o.fill_(1.0), reps=5) / 1e9, 2))
    o4 = o[:src.numel()]
    copies.append(round(2 * o4.numel() * 4 / timeit(lambda: o4.copy_(src), reps=5) / 1e9, 2))
print(json.dumps({"tag": "torch fill", "TBps_by_region": fills}))
print(json.dumps({"tag": "torch copy (4 GB, read from the buffer end)", "TBps_by_region": copies}))
# 8-bit and 1-bit decodes of the same output regions
for bps, coder in ((8, _lib.CODER_INT), (1, _lib.CODER_VDIF)):
    pl = 8000 * bps // 2
    st2 = pl + 32
    a2 = nfr * (st2 + 128000)
    r = []
    for f0 in regions:
        o = out[f0 * per:(f0 + nfr) * per]
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, pl, coder, bps, src0=32, src_stride=st2, out=o), reps=5)
        r.append(round(a2 / ms / 1e9, 2))
    print(json.dumps({"tag": "%d-bit decode, same output regions" % bps, "TBps_by_region": r}), flush=True)
