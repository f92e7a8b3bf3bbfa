#!/usr/bin/env python3
"""Experiment: does the per-frame structure cost the flat 2-bit decode anything?
Same kernel, same bytes, frames of different payload sizes (no headers)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib          # noqa: E402
from tools.bench_formats import timeit          # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
nbytes = int(gib * 2 ** 30)
kernels.init()
variants = [int(v) for v in os.environ.get('BB_VARIANT', '3').split(',')]
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device='cuda')
out = torch.empty(nbytes * 4, dtype=torch.float32, device='cuda')
cases = ((8000, 32), (8000, 0), (8192, 0), (16384, 0), (65536, 0), (160000, 0),
         (1 << 20, 0), (1 << 24, 0), (nbytes, 0))
if os.environ.get('BB_CASES') == 'short':
    cases = ((8000, 32), (8000, 0), (8192, 0), (10000, 16))
for payload, header, variant in [(p, h, v) for p, h in cases for v in variants]:
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    stride = payload + header
    nfr = nbytes // stride
    o = out[:nfr * payload * 4]
    for indexed in (False, True):
        src = None
        if indexed:
            src = (torch.arange(nfr, device='cuda', dtype=torch.int64) * stride + header)
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2,
                                                  src=src, src0=header, src_stride=stride, out=o))
        alg = nfr * (stride + payload * 16)
        print(json.dumps(dict(variant=variant, payload=payload, header=header, indexed=indexed, nframes=nfr,
                              ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
