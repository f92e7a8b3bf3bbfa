#!/usr/bin/env python3
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
dev = 'cuda'
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
fn_, pn, nth = 8032, 8000, 8
nsets = nbytes // (fn_ * nth)
perm = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=dev)
pos = torch.arange(nsets, device=dev, dtype=torch.int64)[:, None] * nth + perm[None, :]
src = (pos * fn_ + 32).reshape(-1).contiguous()
out = torch.empty(nsets * nth * pn * 4, dtype=torch.float32, device=dev)
alg = nsets * nth * fn_ + out.numel() * 4
for tpw in (12, 7, 6, 5, 4, 3, 2):
    for blocks in (0, 65536, 262144):
        kernels.tune(_lib.TUNE_TILES_PER_WAVE, tpw)
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src,
                                                  complex_data=True, out=out))
        print(json.dumps(dict(rows_tpw=min(tpw, 8), blocks=blocks, ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
