#!/usr/bin/env python3
"""Aligned flat kernel: workgroups per CU (capped with unused dynamic LDS)
against grid size, mid-size and large launches."""
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
per = payload * 4
for nfr in (1 << 18, 1 << 20):
    alg = nfr * (stride + payload * 16)
    for wg_per_cu, pad in ((8, 0), (7, 23000), (6, 27000), (5, 32000), (4, 40000), (3, 54000), (2, 65536)):
        kernels.tune(_lib.TUNE_LDS_PAD, pad)
        row = dict(frames=nfr, wg_per_cu=wg_per_cu)
        for blocks in (256 * wg_per_cu, 512 * wg_per_cu, 16384, 131072):
            kernels.tune(_lib.TUNE_BLOCKS, blocks)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                                      src_stride=stride, out=out[:nfr * per]), reps=5)
            row['b%d' % blocks] = round(alg / ms / 1e9, 2)
        print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_LDS_PAD, 0); kernels.tune(_lib.TUNE_BLOCKS, 0)
