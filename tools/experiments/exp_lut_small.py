#!/usr/bin/env python3
"""k_decode_flat_lut<.,.,2,4> against <.,.,2,16> for 4-tile work items (same tensors)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
for FN, PN in ((8032, 8000), (10016, 10000)):
    for gib in (8, 2):
        nfr = (gib << 30) // FN
        buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
        out = torch.empty(nfr * PN * 4, dtype=torch.float32, device=dev)
        src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + (FN - PN)
        res = {}
        for name, small in (('2x16', 0), ('2x4', 1), ('2x16 again', 0), ('2x4 again', 1)):
            kernels.tune(_lib.TUNE_LUT_SMALL, small)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, 0, 2, src=src, out=out), reps=8)
            res[name] = round(nfr * (FN + PN * 16) / ms / 1e9, 3)
        kernels.tune(_lib.TUNE_LUT_SMALL, 0)
        print(json.dumps(dict(payload=PN, GiB=gib, kernel=_lib.last_kernel()[:50], TBps=res)), flush=True)
        del buf, out, src
        torch.cuda.empty_cache()
