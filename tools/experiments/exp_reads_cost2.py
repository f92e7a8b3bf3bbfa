#!/usr/bin/env python3
"""What the 6 % of reads cost the headline launch now: the same launch with every
frame's source pointing at ONE frame (input served by the caches) against the real
index, same tensors."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
FN, PN = 8032, 8000
for nfr in ((8 << 30) // FN, (2 << 30) // FN):
    buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
    out = torch.empty(nfr * PN * 4, dtype=torch.float32, device=dev)
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
    one = torch.full((nfr,), 32, device=dev, dtype=torch.int64)
    few = (torch.arange(nfr, device=dev, dtype=torch.int64) % 1024) * FN + 32          # 8 MiB of input, L2 / MALL resident
    res = {}
    w32 = (torch.arange(nfr, device=dev, dtype=torch.int64) % 4096) * FN + 32          # 33 MB of input
    w128 = (torch.arange(nfr, device=dev, dtype=torch.int64) % 16384) * FN + 32        # 131 MB
    w512 = (torch.arange(nfr, device=dev, dtype=torch.int64) % 65536) * FN + 32        # 526 MB (beyond the Infinity Cache)
    for name, s in (('real index', src), ('one frame', one), ('1024 frames', few), ('33 MB window', w32),
                    ('131 MB window', w128), ('526 MB window', w512), ('real index again', src)):
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, 0, 2, src=s, out=out), reps=6)
        res[name] = dict(ms=round(ms, 3), write_TBps=round(nfr * PN * 16 / ms / 1e9, 3), alg_TBps=round(nfr * (FN + PN * 16) / ms / 1e9, 3))
    fill = timeit(lambda: out.fill_(1.0), reps=4)
    res['torch fill_'] = dict(ms=round(fill, 3), write_TBps=round(nfr * PN * 16 / fill / 1e9, 3))
    print(json.dumps(dict(frames=nfr, result=res)), flush=True)
    del buf, out, src, one, few
    torch.cuda.empty_cache()
