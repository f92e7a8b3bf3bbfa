#!/usr/bin/env python3
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device='cuda')
nf1 = nbytes // 8032
out = torch.empty(nf1 * 32000, dtype=torch.float32, device='cuda')
alg = nf1 * 8032 + nf1 * 128000
for variant in (2, 3, 5, 4):
    for blocks in (32768, 65536, 131072, 196608, 262144):
        kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(lambda: kernels.decode_frames(buf, nf1, 8000, 0, 2, src0=32, src_stride=8032, out=out))
        print(json.dumps(dict(variant=variant, blocks=blocks, ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
# 8-bit flat (DADA-like)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
nb = nbytes // 4 * 4
o8 = out[:nb // 1] if out.numel() >= nb else torch.empty(nb, dtype=torch.float32, device='cuda')
for blocks in (16384, 65536, 131072, 262144, 524288):
    kernels.tune(_lib.TUNE_BLOCKS, blocks)
    n8 = min(nb, out.numel())
    ms = timeit(lambda: kernels.decode_frames(buf, 1, n8, _lib.CODER_INT, 8, src0=0, out=out[:n8]))
    print(json.dumps(dict(case='int8 flat', blocks=blocks, ms=round(ms, 3), TBps=round(n8 * 5 / ms / 1e9, 3))), flush=True)
