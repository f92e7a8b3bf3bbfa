#!/usr/bin/env python3
"""Launch-size dependence of the 2-bit flat decode: frames per launch from 2^13
to 2^20, single launches (sync between) and 4 back-to-back launches."""
import json, os, sys
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

def run(nfr, back_to_back, reps=7):
    fn = lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                       src_stride=stride, out=out[:nfr * payload * 4])
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(back_to_back):
            fn()
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) / back_to_back)
    return float(np.median(ts))

for variant in (5, 0):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    for lg in range(13, 21):
        nfr = 1 << lg
        alg = nfr * (stride + payload * 16)
        for b2b in (1, 4):
            ms = run(nfr, b2b)
            print(json.dumps(dict(variant=variant, frames=nfr, GB=round(alg / 1e9, 2), back_to_back=b2b,
                                  ms=round(ms, 4), TBps=round(alg / ms / 1e9, 3))), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
