#!/usr/bin/env python3
"""Pipelined flat kernel: power-of-two grids against odd / prime ones (the
distance between a workgroup's successive items is grid x 64000 bytes)."""
import json, os, sys
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
for lg in (17, 18, 19, 20):
    nfr = 1 << lg
    alg = nfr * (stride + payload * 16)
    row = dict(frames=nfr)
    for blocks in (65536, 65537, 98304, 100003, 120011, 131071, 131072, 131101, 150001, 196613, 262144, 262147):
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                                  src_stride=stride, out=out[:nfr * payload * 4]), reps=5)
        row['b%d' % blocks] = round(alg / ms / 1e9, 2)
    print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_BLOCKS, 0)
