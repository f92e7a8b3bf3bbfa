#!/usr/bin/env python3
"""k_decode_gather_select: payload bytes staged per work item, single-slot
(Mark 5B 16 channels) and 8-thread VDIF selections, 4 GiB inputs in HBM."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
nbytes = 4 << 30
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
out = torch.empty(17 << 30, dtype=torch.float32, device=dev)
nfr = nbytes // 10016
src = torch.arange(nfr, device=dev, dtype=torch.int64) * 10016 + 16
fn, pn, nth = 8032, 8000, 8
nsets = nbytes // (fn * nth)
src8 = (torch.arange(nsets * nth, device=dev, dtype=torch.int64) * fn + 32)
for stage in (1024, 2048, 4096, 8192, 16384, 32768):
    kernels.tune(_lib.TUNE_SELECT_BYTES, stage)
    res = {}
    for sel in ([1, 6], list(range(8)), list(range(16))):
        w = torch.tensor(sel, dtype=torch.int32, device=dev)
        n = nfr * 2500 * len(sel)
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, 10000, _lib.CODER_MARK5B, 2, chunk=16, src=src,
                                                  out=out[:n], within=w), reps=3)
        res['m5b %d/16' % len(sel)] = round((nfr * 10016 + n * 4) / ms / 1e9, 2)
    for chans in ([3], [3, 4, 5, 6], list(range(16))):
        sel = [2 * c + k for c in chans for k in (0, 1)]
        w = torch.tensor(sel, dtype=torch.int32, device=dev)
        n = nsets * 1000 * nth * len(sel)
        ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src8,
                                                  complex_data=True, out=out[:n], within=w), reps=3)
        res['vdif8 %d/16' % len(chans)] = round((nsets * nth * fn + n * 4) / ms / 1e9, 2)
    print(json.dumps(dict(stage_bytes=stage, kernel=_lib.last_kernel(), TBps_moved=res)), flush=True)
