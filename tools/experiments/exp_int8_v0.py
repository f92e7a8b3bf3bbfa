#!/usr/bin/env python3
"""8-bit flat decode: non-persistent one-workgroup-per-item kernel (variant 0)
against the persistent pipelined one (variant 5), several payload sizes."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nbytes, dtype=torch.float32, device='cuda')
for coder, name in ((_lib.CODER_INT, 'int8'), (_lib.CODER_VDIF, 'vdif8')):
    for payload, header in ((128 << 20, 4096), (8000, 32), (10000, 16), (65536, 0), (1 << 20, 0)):
        stride = payload + header
        nfr = min(nbytes // stride, out.numel() // payload)
        alg = nfr * (stride + payload * 4)
        for variant, blocks in ((5, 0), (0, 0), (0, 1 << 20), (0, 1 << 18), (5, 0), (0, 0)):
            kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
            kernels.tune(_lib.TUNE_BLOCKS, blocks)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, coder, 8, src0=header,
                                                      src_stride=stride, out=out[:nfr * payload]))
            print(json.dumps(dict(coder=name, payload=payload, variant=variant, blocks=blocks,
                                  ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
