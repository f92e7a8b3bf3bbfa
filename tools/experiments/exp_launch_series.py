#!/usr/bin/env python3
"""Time series of identical mid-size launches, then randomised offsets."""
import json, os, sys, random
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
kernels.init()
payload, header = 8000, 32
stride = payload + header
nmax = 1 << 20
buf = torch.randint(0, 256, (nmax * stride + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nmax * payload * 4, dtype=torch.float32, device='cuda')
per = payload * 4

def series(nfr, f0, n):
    o = out[f0 * per:(f0 + nfr) * per]
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header + f0 * stride,
                              src_stride=stride, out=o)
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]

for nfr in (1 << 17, 1 << 19):
    alg = nfr * (stride + payload * 16)
    ts = series(nfr, 0, 60)
    print(json.dumps(dict(frames=nfr, first=0, TBps_series=[round(alg / t / 1e9, 2) for t in ts])), flush=True)
random.seed(1)
nfr = 1 << 17
alg = nfr * (stride + payload * 16)
offs = [k * nfr for k in range(8)] * 3
random.shuffle(offs)
res = {}
for f0 in offs:
    ts = series(nfr, f0, 5)
    res.setdefault(f0, []).append(round(alg / float(np.median(ts)) / 1e9, 2))
print(json.dumps({"by_offset_2^17": res}), flush=True)
# virtual address of the buffers
print(json.dumps(dict(out_ptr=hex(out.data_ptr()), buf_ptr=hex(buf.data_ptr()))))
