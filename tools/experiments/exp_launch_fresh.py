#!/usr/bin/env python3
"""Launch sizes with FRESH, exactly sized output tensors (what a reader's
read() allocates), kernel choices side by side; repeated to see the scatter."""
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
for rep in range(2):
    for lg in (15, 16, 17, 18, 19, 20):
        nfr = 1 << lg
        alg = nfr * (stride + payload * 16)
        row = dict(frames=nfr, rep=rep)
        for name, variant, blocks in (('pipelined', 5, 131072), ('plain', 0, 0), ('auto', 5, 0)):
            o = torch.empty(nfr * payload * 4, dtype=torch.float32, device='cuda')
            kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
            kernels.tune(_lib.TUNE_BLOCKS, blocks)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                                      src_stride=stride, out=o), reps=5)
            row[name] = round(alg / ms / 1e9, 2)
            del o
            torch.cuda.empty_cache()
        print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_BLOCKS, 0)
