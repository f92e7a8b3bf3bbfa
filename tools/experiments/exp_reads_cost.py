#!/usr/bin/env python3
"""Does the 6 % of HBM reads cost the mid-size decode its 20 %?  Same launch
with every frame reading the SAME payload (src_stride = 0: input served by the
caches, writes unchanged) against the normal strided input."""
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
for lg in (16, 17, 18, 20):
    nfr = 1 << lg
    o = out[:nfr * payload * 4]
    row = dict(frames=nfr)
    for variant, vname in ((5, 'pipelined'), (0, 'plain')):
        kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
        kernels.tune(_lib.TUNE_BLOCKS, 131072 if variant == 5 else 0)
        for sname, st in (('hbm_input', stride), ('cached_input', 0)):
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                                      src_stride=st, out=o), reps=5)
            row['%s_%s' % (vname, sname)] = round(nfr * payload * 16 / ms / 1e9, 2)   # output bytes only
    ms = timeit(lambda: o.fill_(1.0), reps=5)
    row['torch_fill'] = round(nfr * payload * 16 / ms / 1e9, 2)
    print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_BLOCKS, 0)
