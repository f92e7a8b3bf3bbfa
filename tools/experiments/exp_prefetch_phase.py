"""Does reading a chunk of the file right BEFORE its decode launch (so that the
decode's loads might be served by the memory-side cache and HBM sees stores
only) beat the plain launch?  cfg2 geometry, chunks of 2^14 / 2^13 frames.
    python tools/experiments/exp_prefetch_phase.py
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402
from baseband_amd._lib import lib               # noqa: E402
from baseband_amd.kernels import _ptr, _stream  # noqa: E402

kernels.init()
dev = torch.device('cuda', 0)
FB, PB = 8032, 8000
nfr_total = 1 << 20
img = torch.empty(nfr_total * FB + 256, dtype=torch.uint8, device=dev)
for lo in range(0, img.numel(), 1 << 30):
    img[lo:lo + (1 << 30)].random_(0, 256)
out = torch.empty(nfr_total * PB * 4, dtype=torch.float32, device=dev)
offs = torch.empty(1 << 20, dtype=torch.int64, device=dev)
count = torch.zeros(1, dtype=torch.int64, device=dev)


def decode(f0, nf):
    kernels.decode_frames(img[f0 * FB:], nf, PB, _lib.CODER_VDIF, 2, src0=32, src_stride=FB,
                          out=out[f0 * PB * 4:(f0 + nf) * PB * 4])


def touch(f0, nf):
    chunk = img[f0 * FB:(f0 + nf) * FB]
    lib.bb_mark5b_locate(_ptr(chunk), chunk.numel(), _ptr(offs), offs.numel(), _ptr(count), _stream(chunk))


def timed(fn, reps=3):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return min(ts)


whole = timed(lambda: decode(0, nfr_total))
print(json.dumps({"case": "one launch", "ms": round(whole, 3)}), flush=True)
for lchunk in (14, 13, 15):
    nf = 1 << lchunk
    plain = timed(lambda: [decode(f0, nf) for f0 in range(0, nfr_total, nf)])
    sweep = timed(lambda: [touch(f0, nf) for f0 in range(0, nfr_total, nf)])
    both = timed(lambda: [(touch(f0, nf), decode(f0, nf)) for f0 in range(0, nfr_total, nf)])
    print(json.dumps({"chunk_frames": nf, "chunk_MiB": round(nf * FB / 2 ** 20, 1), "launches": nfr_total // nf,
                      "decode_chunks_ms": round(plain, 3), "touch_chunks_ms": round(sweep, 3),
                      "touch_then_decode_ms": round(both, 3)}), flush=True)
