#!/usr/bin/env python3
"""One 2^20-frame launch against the same work as 2, 4, 8, 16 launches over
consecutive regions (same process, same buffers)."""
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
alg = nmax * (stride + payload * 16)
for variant in (5, 0):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    for parts in (1, 2, 4, 8, 16, 1):
        nfr = nmax // parts
        def fn():
            for k in range(parts):
                f0 = k * nfr
                kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header + f0 * stride,
                                      src_stride=stride, out=out[f0 * per:(f0 + nfr) * per])
        ms = timeit(fn, reps=5)
        print(json.dumps(dict(variant=variant, launches=parts, ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
