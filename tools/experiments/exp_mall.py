"""Does the memory-side cache (256 MiB "infinity cache") serve repeated reads?
k_encode_flat (a streaming read: 16 B of float32 in per byte out at 2 bits)
over the same N MiB again and again, N = 16 .. 2048; then the same with 2 GiB
of decode output (nt stores) written between two reads of the buffer.
    python tools/experiments/exp_mall.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402

kernels.init()
x = torch.empty(2 << 28, dtype=torch.float32, device='cuda').normal_(0, 2.2)      # 2 GiB


def rate(n_mib, reps=40, between=None):
    n = n_mib << 18
    for _ in range(3):
        kernels.encode_flat(x[:n], 0, 2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if between is None:
        a.record()
        for _ in range(reps):
            kernels.encode_flat(x[:n], 0, 2)
        b.record(); b.synchronize()
        return n * 4 * reps / a.elapsed_time(b) / 1e9
    tot = 0.0
    for _ in range(reps):
        between()
        a.record()
        kernels.encode_flat(x[:n], 0, 2)
        b.record(); b.synchronize()
        tot += a.elapsed_time(b)
    return n * 4 * reps / tot / 1e9


for n_mib in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    print(json.dumps({"read_MiB": n_mib, "back_to_back_TBps": round(rate(n_mib), 2)}), flush=True)

# the same read after (a) nothing (launch alone), (b) 2 GiB of nt decode stores, (c) a plain fill of 2 GiB
raw = torch.randint(0, 256, (128 << 20,), dtype=torch.uint8, device='cuda')
big = torch.empty(512 << 20, dtype=torch.float32, device='cuda')
dec = lambda: kernels.decode_frames(raw, (128 << 20) // 8192, 8192, _lib.CODER_VDIF, 2, src0=0, src_stride=8192, out=big)
for n_mib in (32, 64, 128):
    print(json.dumps({"read_MiB": n_mib,
                      "alone_TBps": round(rate(n_mib, reps=10, between=lambda: torch.cuda.synchronize()), 2),
                      "after_2GiB_written_TBps": round(rate(n_mib, reps=10, between=lambda: (dec(), None)[1]), 2)}), flush=True)
