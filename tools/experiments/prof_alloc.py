#!/usr/bin/env python3
"""The allocation sequence of tools/experiments/exp_alloc.py with THREE decode launches per
size (one untimed, two timed with HIP events), for counter passes under
rocprofv3: which hardware counters differ between a launch whose output
landed well (6.4-6.8 TB/s) and one whose output landed badly (5.3)?"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
kernels.init()
dev = torch.device('cuda')
FN, PN, SPF = 8032, 8000, 32000
for lg in (15, 16, 17, 18, 19, 20):
    nfr = 1 << lg
    buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
    out = torch.empty(nfr * SPF, dtype=torch.float32, device=dev)
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
    ms = []
    for k in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        kernels.decode_frames(buf, nfr, PN, _lib.CODER_VDIF, 2, src=src, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print(json.dumps(dict(frames=nfr, out_ptr=hex(out.data_ptr()), in_ptr=hex(buf.data_ptr()),
                          TBps=[round(nfr * (FN + SPF * 4) / m / 1e9, 3) for m in ms])), flush=True)
    del buf, out, src
    torch.cuda.empty_cache()
