#!/usr/bin/env python3
"""Flat decode, every coder / bits-per-sample: default kernel choice against
the plain kernel (variant 0) and the pipelined ones (3, 5)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
out_elems = 32 << 30 >> 2            # 32 GiB of float32 output
buf = torch.randint(0, 256, ((8 << 30) + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(out_elems, dtype=torch.float32, device='cuda')
payload, header = 8000, 32
stride = payload + header
for coder, name, bpss in ((_lib.CODER_VDIF, 'vdif', (1, 2, 4, 8)), (_lib.CODER_MARK5B, 'mark5b', (1, 2)),
                          (_lib.CODER_INT, 'int', (4, 8))):
    for bps in bpss:
        per = payload * 8 // bps
        nfr = min((8 << 30) // stride, out_elems // per)
        alg = nfr * (stride + per * 4)
        for variant in (5, 0, 3, 5, 0):
            kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
            if bps == 8:
                kernels.tune(_lib.TUNE_TILES_PER_WAVE_8BIT, 12 if variant != 3 else 12)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, coder, bps, src0=header,
                                                      src_stride=stride, out=out[:nfr * per]))
            print(json.dumps(dict(coder=name, bps=bps, variant=variant, ms=round(ms, 3),
                                  TBps=round(alg / ms / 1e9, 3))), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
