"""Does it matter where the INPUT lies?  cfg2 launches of 2^15 / 2^16 / 2^18
frames into arena outputs (fresh block per draw), reading the same bytes from
a plain torch tensor and from an arena block, taking turns.
    python tools/experiments/exp_image_arena.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

FRAME, PAYLOAD, HDR = 8032, 8000, 32
dev = torch.device('cuda', 0)
kernels.init()
nframes = (8 << 30) // FRAME
g = torch.Generator(device=dev)
g.manual_seed(1)
ar = arena.Arena(280 << 30)
img_t = torch.empty(nframes * FRAME, dtype=torch.uint8, device=dev)
v = img_t[:nframes * FRAME // 4 * 4].view(torch.int32)
for lo in range(0, v.numel(), 1 << 28):
    hi = min(v.numel(), lo + (1 << 28))
    v[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
img_a = ar.empty(nframes * FRAME, dtype=torch.uint8)
img_a.copy_(img_t)


def rate(img, out, nf, f0):
    ts = []
    for r in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(img, nf, PAYLOAD, _lib.CODER_VDIF, 2, src0=HDR + f0 * FRAME, src_stride=FRAME, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return round(nf * (FRAME + PAYLOAD * 16) / float(np.median(ts)) / 1e9, 3)


for nf in (1 << 15, 1 << 16, 1 << 18):
    res = {"plain": [], "arena": []}
    held = []
    for draw in range(6):
        out = ar.empty(nf * PAYLOAD * 4)
        f0 = (draw * 150001) % (nframes - nf)
        for name, img in (("plain", img_t), ("arena", img_a)) if draw % 2 == 0 else (("arena", img_a), ("plain", img_t)):
            res[name].append(rate(img, out, nf, f0))
        held.append(ar.empty((64 << 20) // 4))
        del out
    del held
    print(json.dumps({"frames": nf, "TBps_image_plain": res["plain"], "TBps_image_in_arena": res["arena"],
                      "median_ratio": round(float(np.median(np.array(res["arena"]) / np.array(res["plain"]))), 4)}), flush=True)
