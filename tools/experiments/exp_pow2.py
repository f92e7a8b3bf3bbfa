#!/usr/bin/env python3
"""Are launches of exactly 2^k frames special?  With the striped work order the
write fronts of a launch lie (frames / stripes) x (bytes per frame) apart: for
2^k frames of 128000 output bytes that is a multiple of 16 MiB or more, so all
fronts sit at the same place of any power-of-two address interleave.  Same
process, fresh tensors per case, frame counts 2^k against neighbours that are
not, stripes on (default) and off."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
FN, PN, SPF = 8032, 8000, 32000
for rep in range(2):
    for nfr in (65536, 65521, 70001, 131072, 131101, 120011, 262144, 262147, 250007, 524288, 524309, 500009):
        buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
        out = torch.empty(nfr * SPF, dtype=torch.float32, device=dev)
        src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
        res = {}
        for name, lw in (('auto', -1), ('file order', 0), ('4 stripes', 2), ('16 stripes', 4)):
            kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
            ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, _lib.CODER_VDIF, 2, src=src, out=out), reps=5)
            res[name] = round(nfr * (FN + SPF * 4) / ms / 1e9, 3)
        kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
        print(json.dumps(dict(rep=rep, frames=nfr, pow2=(nfr & (nfr - 1)) == 0, TBps=res)), flush=True)
        del buf, out, src
        torch.cuda.empty_cache()
