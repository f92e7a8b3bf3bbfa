"""In memory made of physically contiguous 1 GiB handles (the deterministic
slow case, 5.30 TB/s at 2^16 cfg2 frames: profiles/r03d_arena_probe4.log) --
does the rate depend on the SPACING of the launch's write fronts (stripes of
the work order), i.e. on the number of frames and of stripes?
    BB_ARENA_CHUNK_MIB=1024 BB_ARENA_TRIES=1 python tools/experiments/exp_front_spacing.py
"""
import json
import os
import sys

os.environ.setdefault('BB_ARENA_CHUNK_MIB', '1024')
os.environ.setdefault('BB_ARENA_TRIES', '1')
os.environ.setdefault('BB_ARENA_STEP_GIB', '48')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
F, P = 8032, 8000
nimg = (4 << 30) // F
image = torch.randint(0, 256, (nimg * F,), dtype=torch.uint8, device=dev)
ar = arena.Arena(100 << 30)
out = ar.empty(80000 * 32000)             # 10.2 GB: ten contiguous 1 GiB handles


def rate(nf, k=0):
    ts = []
    for r in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0 = ((k * 4 + r) * 70001) % (nimg - nf)
        a.record()
        kernels.decode_frames(image, nf, P, _lib.CODER_VDIF, 2, src0=32 + f0 * F, src_stride=F, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return round(nf * (F + P * 16) / float(np.median(ts)) / 1e9, 3)


print(json.dumps({"arena": {k: ar.stats()[k] for k in ("chunk_bytes", "steps", "bytes_backed")}}), flush=True)
res = {}
for nf in list(range(60000, 72001, 1000)) + [65536, 32768, 49152, 16384]:
    res[nf] = rate(nf)
print(json.dumps({"default stripes, TBps by frames": res}), flush=True)
for nf in (65536, 61000):
    row = {}
    for lw in (0, 1, 2, 3, 4, 5, 6, 8):
        kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
        row[1 << lw] = rate(nf, lw)
    kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
    print(json.dumps({"frames": nf, "TBps by stripes": row}), flush=True)
