#!/usr/bin/env python3
"""Is the mid-size dip a matter of WHERE the output lies?  Same launch
(2^17 and 2^18 frames) at different offsets of the big output / input buffers,
and a pure fill of the same output ranges."""
import json, os, sys
import numpy as np
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
per = payload * 4
for nfr in (1 << 16, 1 << 17, 1 << 18, 1 << 19):
    alg = nfr * (stride + payload * 16)
    for k in range(0, nmax // nfr, max(1, nmax // nfr // 4)):
        f0 = k * nfr
        o = out[f0 * per:(f0 + nfr) * per]
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2,
                                                  src0=header + f0 * stride, src_stride=stride, out=o))
        ms_in0 = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2,
                                                      src0=header, src_stride=stride, out=o))
        ms_fill = timeit(lambda: o.fill_(1.0))
        print(json.dumps(dict(frames=nfr, first_frame=f0, out_GB=round(o.numel() * 4 / 1e9, 1),
                              decode_TBps=round(alg / ms / 1e9, 2),
                              decode_input_at_0_TBps=round(alg / ms_in0 / 1e9, 2),
                              fill_TBps=round(o.numel() * 4 / ms_fill / 1e9, 2))), flush=True)
# fresh, exactly-sized allocations
del out
torch.cuda.empty_cache()
for nfr in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20):
    alg = nfr * (stride + payload * 16)
    o = torch.empty(nfr * per, dtype=torch.float32, device='cuda')
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2,
                                              src0=header, src_stride=stride, out=o))
    ms_fill = timeit(lambda: o.fill_(1.0))
    print(json.dumps(dict(fresh_alloc_frames=nfr, decode_TBps=round(alg / ms / 1e9, 2),
                          fill_TBps=round(o.numel() * 4 / ms_fill / 1e9, 2))), flush=True)
    del o
    torch.cuda.empty_cache()
