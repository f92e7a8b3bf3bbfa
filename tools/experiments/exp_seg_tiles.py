#!/usr/bin/env python3
"""Plain flat kernel: tiles per workgroup (work item size) against launch size."""
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
kernels.tune(_lib.TUNE_FLAT_VARIANT, 0)
for lg in (15, 17, 18, 20):
    nfr = 1 << lg
    o = out[:nfr * payload * 4]
    for sname, st in (('hbm_input', stride), ('cached_input', 0)):
        row = dict(frames=nfr, input=sname)
        for seg in (1, 2, 4, 8, 16, 32, 64, 128):
            kernels.tune(_lib.TUNE_SEG_TILES, seg)
            try:
                ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                                          src_stride=st, out=o), reps=5)
                row['seg%d' % seg] = round(nfr * payload * 16 / ms / 1e9, 2)
            except Exception as exc:
                row['seg%d' % seg] = str(exc)[-40:]
        print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_SEG_TILES, 32)
