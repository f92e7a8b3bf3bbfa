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
out = torch.empty(nbytes * 4, dtype=torch.float32, device='cuda')
for payload, header in ((10000, 16), (8000, 32), (5000, 32), (65536, 0)):
    stride = payload + header
    nfr = nbytes // stride
    alg = nfr * (stride + payload * 16)
    for tpw in (16, 12, 10, 8, 7, 6):
        for blocks in (0, 262144):
            kernels.tune(_lib.TUNE_TILES_PER_WAVE, tpw)
            kernels.tune(_lib.TUNE_BLOCKS, blocks)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, 0, 2, src0=header, src_stride=stride,
                                                      out=out[:nfr * payload * 4]))
            print(json.dumps(dict(payload=payload, tpw_max=tpw, blocks=blocks, ms=round(ms, 3),
                                  TBps=round(alg / ms / 1e9, 3))), flush=True)
