"""The headline launch (cfg2, 8 GiB in, 127.5 GiB out) with the image and the
output each in a plain torch tensor or in an arena block -- ONE combination per
process, as bench.py meets it (tools/experiments/exp_headline_arena.py has all four in one
process, where they share the device's memory).
    python tools/experiments/exp_headline_alloc.py plain|arena plain|arena
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
iw, ow = sys.argv[1], sys.argv[2]
order = sys.argv[3] if len(sys.argv) > 3 else 'image_first'
ar = arena.Arena(280 << 30) if 'arena' in (iw, ow) else None


def make(which, n, dtype):
    return ar.empty(n, dtype=dtype) if which == 'arena' else torch.empty(n, dtype=dtype, device=dev)


if order == 'image_first':
    img = make(iw, nframes * FRAME, torch.uint8)
    out = make(ow, nframes * PAYLOAD * 4, torch.float32)
else:
    out = make(ow, nframes * PAYLOAD * 4, torch.float32)
    img = make(iw, nframes * FRAME, torch.uint8)
v = img[:nframes * FRAME // 4 * 4].view(torch.int32)
for lo in range(0, v.numel(), 1 << 28):
    hi = min(v.numel(), lo + (1 << 28))
    v[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
alg = nframes * (FRAME + PAYLOAD * 16)
ts = []
for r in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    kernels.decode_frames(img, nframes, PAYLOAD, _lib.CODER_VDIF, 2, src0=HDR, src_stride=FRAME, out=out)
    b.record()
    b.synchronize()
    if r >= 2:
        ts.append(a.elapsed_time(b))
print(json.dumps({"image": iw, "output": ow, "order": order, "frac": round(alg / float(np.median(ts)) / 1e6 / 8000, 4),
                  "ms": [round(t, 2) for t in ts],
                  "arena": None if ar is None else {k: ar.stats()[k] for k in ('steps', 'probes', 'last_probe_gbps')}}), flush=True)
