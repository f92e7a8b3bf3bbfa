"""Does reading a chunk of the input in one burst (so that it sits in the
256 MiB Infinity Cache) and then decoding that chunk beat one launch that
reads while it writes?  cfg2 headline (8 GiB in, 127.5 GiB out): the whole
file in ONE launch, in chunks without a prefetch, and in chunks with a
read burst (torch.sum of the chunk's bytes) right before each chunk's decode.
    python tools/experiments/exp_mall_prefetch.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402
import bench                                    # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
F, P = 8032, 8000
nframes = (8 << 30) // F
image, _ = bench.image_buffer(nframes * F, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
out = torch.empty(nframes * 32000, dtype=torch.float32, device=dev)
alg = nframes * (F + P * 16)
sink = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(fn, reps=4):
    ts = []
    for r in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def whole():
    kernels.decode_frames(image, nframes, P, _lib.CODER_VDIF, 2, src0=32, src_stride=F, out=out)


def chunked(cf, prefetch):
    def run():
        for f0 in range(0, nframes, cf):
            n = min(cf, nframes - f0)
            if prefetch:
                w = image[f0 * F:(f0 + n) * F]
                sink.add_(w[:w.numel() // 8 * 8].view(torch.int64).sum())
            kernels.decode_frames(image, n, P, _lib.CODER_VDIF, 2, src0=32 + f0 * F, src_stride=F,
                                  out=out[f0 * 32000:(f0 + n) * 32000])
    return run


ms = timed(whole)
print(json.dumps({"case": "one launch", "ms": round(ms, 3), "frac": round(alg / ms / 1e6 / 8000, 4)}), flush=True)
for cf in (1 << 12, 1 << 13, 1 << 14, 24000, 1 << 15):
    row = {"chunk_frames": cf, "chunk_input_MB": round(cf * F / 1e6, 1)}
    for pf in (False, True):
        ms = timed(chunked(cf, pf))
        row["prefetch" if pf else "no_prefetch"] = {"ms": round(ms, 3), "frac": round(alg / ms / 1e6 / 8000, 4)}
    print(json.dumps(row), flush=True)
# the read bursts alone
def bursts(cf):
    def run():
        for f0 in range(0, nframes, cf):
            n = min(cf, nframes - f0)
            w = image[f0 * F:(f0 + n) * F]
            sink.add_(w[:w.numel() // 8 * 8].view(torch.int64).sum())
    return run
print(json.dumps({"case": "read bursts alone, 2^14 frames each", "ms": round(timed(bursts(1 << 14)), 3)}), flush=True)
