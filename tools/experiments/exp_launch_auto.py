#!/usr/bin/env python3
"""Default kernel choice against forced variants over launch sizes."""
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
for lg in range(13, 21):
    nfr = 1 << lg
    alg = nfr * (stride + payload * 16)
    row = dict(frames=nfr)
    for name, variant, blocks in (('auto', 5, 0), ('pipelined', 5, 131072), ('plain', 0, 0)):
        kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                                  src_stride=stride, out=out[:nfr * payload * 4]), reps=5)
        row[name] = round(alg / ms / 1e9, 2)
    print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_BLOCKS, 0)
