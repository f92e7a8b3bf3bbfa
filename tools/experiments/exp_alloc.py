#!/usr/bin/env python3
"""Where torch puts an output: the caching allocator's default (one hipMalloc
per large tensor) against expandable segments (HIP virtual memory: physical
chunks of 20 MiB mapped side by side), for decode launches of 2^15 .. 2^20
cfg2 frames.  Run once per setting (fresh process each):
    python tools/experiments/exp_alloc.py
    PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python tools/experiments/exp_alloc.py
"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
conf = os.environ.get('PYTORCH_HIP_ALLOC_CONF') or os.environ.get('PYTORCH_CUDA_ALLOC_CONF') or 'default'
FN, PN, SPF = 8032, 8000, 32000
for rep in range(2):
    for lg in (15, 16, 17, 18, 19, 20):
        nfr = 1 << lg
        buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
        out = torch.empty(nfr * SPF, dtype=torch.float32, device=dev)
        src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
        ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, _lib.CODER_VDIF, 2, src=src, out=out), reps=5)
        fill_ms = timeit(lambda: out.fill_(1.0), reps=3)
        print(json.dumps(dict(alloc=conf, rep=rep, frames=nfr, out_GB=round(nfr * SPF * 4 / 1e9, 2),
                              decode_TBps=round(nfr * (FN + SPF * 4) / ms / 1e9, 3),
                              fill_TBps=round(nfr * SPF * 4 / fill_ms / 1e9, 3),
                              kernel=_lib.last_kernel().split(' grid')[0])), flush=True)
        del buf, out, src
        torch.cuda.empty_cache()
