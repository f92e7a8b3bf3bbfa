#!/usr/bin/env python3
"""8-bit flat decode (DADA / GUPPI real / VDIF 8-bit): tiles per wave sweep."""
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
    for payload, header in ((128 << 20, 4096), (8000, 32), (65536, 0)):
        stride = payload + header
        nfr = min(nbytes // stride, out.numel() // payload)
        alg = nfr * (stride + payload * 4)
        for tpw in (8, 12, 16, 20, 24, 32):
            kernels.tune(_lib.TUNE_TILES_PER_WAVE_8BIT, tpw)
            if tpw <= 16:
                kernels.tune(_lib.TUNE_TILES_PER_WAVE, tpw)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, coder, 8, src0=header,
                                                      src_stride=stride, out=out[:nfr * payload]))
            print(json.dumps(dict(coder=name, payload=payload, tpw_max=tpw, ms=round(ms, 3),
                                  TBps=round(alg / ms / 1e9, 3))), flush=True)
        kernels.tune(_lib.TUNE_TILES_PER_WAVE, 12)
