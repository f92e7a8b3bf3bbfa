"""Does the rate of an arena block depend on what ELSE the process has done with
device memory?  (tools/experiments/exp_lds.py had arena blocks at 5.3 TB/s in a process that
never allocated anything large besides; bench.py's mid_size leg, which runs
after the 127.5 GiB headline output was allocated and freed, had them at
6.5-6.7.)  The SAME block is timed before and after the process allocates,
fills and frees a large plain tensor; then new blocks.
    python tools/experiments/exp_arena_history.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

FRAME, PAYLOAD, HDR = 8032, 8000, 32
IMG_FRAMES = 1 << 20
dev = torch.device('cuda', 0)
kernels.init()
g = torch.Generator(device=dev)
g.manual_seed(1)
image = torch.empty(IMG_FRAMES * FRAME // 4, dtype=torch.int32, device=dev)
for lo in range(0, image.numel(), 1 << 28):
    hi = min(image.numel(), lo + (1 << 28))
    image[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
image = image.view(torch.uint8)
torch.cuda.synchronize()
nxt = [0]


def rate(out, nf, reps=6):
    ts = []
    for r in range(reps + 1):
        if nxt[0] + nf > IMG_FRAMES:
            nxt[0] = 0
        first = nxt[0]
        nxt[0] += nf
        win = image[first * FRAME:(first + nf) * FRAME]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, PAYLOAD, _lib.CODER_VDIF, 2, src0=HDR, src_stride=FRAME, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return round(nf * (FRAME + PAYLOAD * 16) / float(np.median(ts)) / 1e9, 3)


def say(**kw):
    kw['free_GiB'] = round(torch.cuda.mem_get_info()[0] / 2 ** 30, 1)
    print(json.dumps(kw), flush=True)


nf = 1 << 16
n = nf * PAYLOAD * 4
ar = arena.Arena(250 << 30)
a1 = ar.empty(n)
say(step="fresh process: first arena block", block="a1", TBps=rate(a1, nf), arena=ar.stats()['bytes_backed'] >> 30)
t1 = torch.empty(n, dtype=torch.float32, device=dev)
say(step="a plain tensor of the same size", block="t1", TBps=rate(t1, nf))
say(step="a1 again", block="a1", TBps=rate(a1, nf))
a2 = ar.empty(n)
say(step="second arena block", block="a2", TBps=rate(a2, nf), arena=ar.stats()['bytes_backed'] >> 30)
big = torch.empty(100 << 28, dtype=torch.float32, device=dev)          # 100 GiB
say(step="100 GiB plain tensor allocated (untouched)", block="a1", TBps=rate(a1, nf))
say(step="...", block="a2", TBps=rate(a2, nf))
say(step="...", block="t1", TBps=rate(t1, nf))
big.fill_(1.0)
torch.cuda.synchronize()
say(step="100 GiB tensor filled", block="a1", TBps=rate(a1, nf))
say(step="...", block="a2", TBps=rate(a2, nf))
a3 = ar.empty(n)
say(step="third arena block, created while the 100 GiB are held", block="a3", TBps=rate(a3, nf), arena=ar.stats()['bytes_backed'] >> 30)
del big
torch.cuda.empty_cache()
say(step="100 GiB tensor freed (back to the driver)", block="a1", TBps=rate(a1, nf))
say(step="...", block="a2", TBps=rate(a2, nf))
say(step="...", block="a3", TBps=rate(a3, nf))
say(step="...", block="t1", TBps=rate(t1, nf))
a4 = ar.empty(n)
say(step="fourth arena block, created after the free", block="a4", TBps=rate(a4, nf), arena=ar.stats()['bytes_backed'] >> 30)
t2 = torch.empty(n, dtype=torch.float32, device=dev)
say(step="a new plain tensor", block="t2", TBps=rate(t2, nf))
# an arena that is backed in ONE step of 48 GiB (what tools/experiments/exp_arena.py measured in r03f-r03h)
ar2 = arena.Arena(48 << 30)
whole = ar2.empty((48 << 30) // 4)
del whole
b1 = ar2.empty(n)
say(step="block of an arena backed in one 48 GiB step", block="b1", TBps=rate(b1, nf), arena=ar2.stats()['bytes_backed'] >> 30)
b2 = ar2.empty(n)
say(step="its second block", block="b2", TBps=rate(b2, nf))
for name, t in (("a1", a1), ("a2", a2), ("a3", a3), ("a4", a4), ("t1", t1), ("t2", t2), ("b1", b1)):
    say(step="at the end", block=name, TBps=rate(t, nf))
