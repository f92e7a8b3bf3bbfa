"""Two outputs of 2^16 cfg2 frames in one process -- `a1`: the first 8 GiB the
driver hands out (an unprobed 8 GiB arena step; 5.3 TB/s in profiles/r03k), `a3`:
an 8 GiB step taken while a 100 GiB tensor is held (6.5) -- decoded three times
each, for counter passes (rocprofv3 --pmc ..., program directly after `--`)."""
import json
import os
import sys

os.environ['BB_ARENA_TRIES'] = '1'
os.environ['BB_ARENA_STEP_GIB'] = '8'
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

FRAME, PAYLOAD, HDR = 8032, 8000, 32
dev = torch.device('cuda', 0)
kernels.init()
nf = 1 << 16
image = torch.randint(0, 256, (4 * nf * FRAME,), dtype=torch.uint8, device=dev)
ar = arena.Arena(250 << 30)
n = nf * PAYLOAD * 4
a1 = ar.empty(n)
big = torch.empty(100 << 28, dtype=torch.float32, device=dev)
a3 = ar.empty(n)
del big
for name, out in (("a1", a1), ("a3", a3), ("a1", a1), ("a3", a3)):
    ts = []
    for r in range(3):
        win = image[r * nf * FRAME:(r + 1) * nf * FRAME]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, PAYLOAD, _lib.CODER_VDIF, 2, src0=HDR, src_stride=FRAME, out=out)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    print(json.dumps({"block": name, "TBps": [round(nf * (FRAME + PAYLOAD * 16) / t / 1e9, 3) for t in ts]}), flush=True)
