#!/usr/bin/env python3
"""LDS gather kernel: bytes staged per work item against bits per sample."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nbytes * 4, dtype=torch.float32, device='cuda')
fn_, pn = 8032, 8000
for bps in (1, 2, 4, 8):
    per = pn * 8 // bps
    for nth in (2, 8, 16):
        nsets = min(nbytes // (fn_ * nth), out.numel() // (nth * per))
        src = (torch.arange(nsets * nth, device='cuda', dtype=torch.int64) * fn_ + 32)
        alg = nsets * nth * (fn_ + per * 4)
        row = dict(bps=bps, threads=nth)
        for gb in (2048, 4096, 8192, 16384, 32768):
            kernels.tune(_lib.TUNE_GATHER_BYTES, gb)
            try:
                ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, bps, chunk=4, nslot=nth, src=src,
                                                          out=out[:nsets * nth * per]), reps=5)
                row['gb%d' % gb] = round(alg / ms / 1e9, 2)
            except Exception as exc:
                row['gb%d' % gb] = type(exc).__name__
        print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_GATHER_BYTES, 8192)
