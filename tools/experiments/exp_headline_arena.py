"""The headline launch (cfg2, 8 GiB in, 127.5 GiB out) into a plain torch.empty
output against an arena block, and with the INPUT image in a plain tensor
against an arena block: same process, taking turns.
    python tools/experiments/exp_headline_arena.py
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


def fill(t):
    v = t.view(torch.int32)
    for lo in range(0, v.numel(), 1 << 28):
        hi = min(v.numel(), lo + (1 << 28))
        v[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)


ar = arena.Arena(280 << 30)
img_t = torch.empty(nframes * FRAME, dtype=torch.uint8, device=dev)
fill(img_t)
img_a = ar.empty(nframes * FRAME, dtype=torch.uint8)
img_a.copy_(img_t)
out_t = torch.empty(nframes * PAYLOAD * 4, dtype=torch.float32, device=dev)
out_a = ar.empty(nframes * PAYLOAD * 4)
print(json.dumps({"arena": ar.stats()}), flush=True)
alg = nframes * (FRAME + PAYLOAD * 16)


def rate(img, out, reps=5):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(img, nframes, PAYLOAD, _lib.CODER_VDIF, 2, src0=HDR, src_stride=FRAME, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return round(alg / float(np.median(ts)) / 1e9, 3)


res = {}
for rnd in range(3):
    for iname, img in (("image plain", img_t), ("image in arena", img_a)):
        for oname, out in (("output plain", out_t), ("output in arena", out_a)):
            res.setdefault(iname + ", " + oname, []).append(rate(img, out))
for k, v in res.items():
    print(json.dumps({"case": k, "TBps": v, "frac": [round(x / 8, 4) for x in v]}), flush=True)
same = bool(torch.equal(out_t.view(torch.int32)[:1 << 28], out_a.view(torch.int32)[:1 << 28]))
print(json.dumps({"outputs_identical_first_GiB": same}))
