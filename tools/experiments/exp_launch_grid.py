#!/usr/bin/env python3
"""2-bit flat decode: grid cap against launch size (frames per launch)."""
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
for lg in range(14, 21):
    nfr = 1 << lg
    alg = nfr * (stride + payload * 16)
    row = dict(frames=nfr, items=2 * nfr)
    for blocks in (4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288):
        if blocks > 4 * nfr:
            continue
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                                  src_stride=stride, out=out[:nfr * payload * 4]))
        row['b%d' % blocks] = round(alg / ms / 1e9, 2)
    print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_BLOCKS, 0)
