#!/usr/bin/env python3
"""Experiment: k_decode_gather staging size / grid."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device='cuda')
fn_, pn, nth = 5032, 5000, 8
nsets = nbytes // (fn_ * nth)
src = (torch.arange(nsets * nth, device='cuda', dtype=torch.int64) * fn_ + 32)
out = torch.empty(nsets * nth * pn * 4, dtype=torch.float32, device='cuda')
alg = nsets * nth * fn_ + out.numel() * 4
for gb in (8192, 16384):
    for blocks in (24576, 32768, 49152, 65536, 131072, 262144):
        kernels.tune(_lib.TUNE_GATHER_BYTES, gb)
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=1, nslot=nth, src=src, out=out))
        print(json.dumps(dict(gather_bytes=gb, blocks=blocks, ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
