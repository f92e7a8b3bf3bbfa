"""16-byte input loads for the flat family.  Same-process A/B
of k_decode_flat_lds (variant 15: dwordx4 loads staged through LDS, k_lds.h)
against the product's k_decode_flat_lut (dword-per-lane loads + ds_bpermute),
outputs in the arena, every launch on the next window of an 8 GiB image.
Needs the experiment build:  BB_EXPERIMENTS=1 python tools/experiments/exp_lds.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

assert _lib.EXPERIMENTS, "run with BB_EXPERIMENTS=1 (make -C baseband_amd/csrc EXPERIMENTS=1)"
dev = torch.device('cuda', 0)
kernels.init()
IMG = 8 << 30
g = torch.Generator(device=dev)
g.manual_seed(1)
image = torch.empty(IMG // 4, dtype=torch.int32, device=dev)
for lo in range(0, image.numel(), 1 << 28):
    hi = min(image.numel(), lo + (1 << 28))
    image[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
image = image.view(torch.uint8)
ar = arena.Arena(200 << 30)
nxt = [0]


def rate(out, nf, frame, payload, hdr, coder, bps, variant, reps=6):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    ts = []
    for r in range(reps + 1):
        if (nxt[0] + nf) * frame > IMG:
            nxt[0] = 0
        first = nxt[0]
        nxt[0] += nf
        win = image[first * frame:(first + nf) * frame]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, payload, coder, bps, src0=hdr, src_stride=frame, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
    return nf * (frame + payload * 32 // bps) / float(np.median(ts)) / 1e9, _lib.last_kernel().split(' ')[0]


cases = [("VDIF 2-bit 8032-byte frames", 8032, 8000, 32, _lib.CODER_VDIF, 2),
         ("Mark 5B 2-bit 10016-byte frames", 10016, 10000, 16, _lib.CODER_MARK5B, 2),
         ("VDIF 1-bit 8032", 8032, 8000, 32, _lib.CODER_VDIF, 1),
         ("VDIF 4-bit 8032", 8032, 8000, 32, _lib.CODER_VDIF, 4),
         ("VDIF 2-bit 8224-byte frames (payload 8192)", 8224, 8192, 32, _lib.CODER_VDIF, 2)]
for name, frame, payload, hdr, coder, bps in cases:
    for lf in (16, 18, 20):
        nf = min(1 << lf, IMG // frame)
        n = nf * payload * 8 // bps
        if n * 4 > (150 << 30):
            nf = (150 << 30) // (payload * 32 // bps)
            n = nf * payload * 8 // bps
        out = ar.empty(n)
        # parity of the two kernels on the same window
        nxt[0] = 0
        kernels.tune(_lib.TUNE_FLAT_VARIANT, 15)
        kernels.decode_frames(image[:nf * frame], nf, payload, coder, bps, src0=hdr, src_stride=frame, out=out)
        k15 = _lib.last_kernel()
        ref = ar.empty(min(n, 1 << 28))
        kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
        m = ref.numel() // (payload * 8 // bps)
        kernels.decode_frames(image[:m * frame], m, payload, coder, bps, src0=hdr, src_stride=frame, out=ref[:m * payload * 8 // bps])
        same = bool(torch.equal(out[:m * payload * 8 // bps].view(torch.int32), ref[:m * payload * 8 // bps].view(torch.int32)))
        del ref
        rows = {5: [], 15: []}
        for rep in range(3):
            for v in (5, 15):
                r, kn = rate(out, nf, frame, payload, hdr, coder, bps, v)
                rows[v].append(round(r, 3))
        print(json.dumps({"case": name, "frames": nf, "output_GB": round(n * 4 / 1e9, 1), "bit_identical": same,
                          "k_decode_flat_lut_TBps": rows[5], "k_decode_flat_lds_TBps": rows[15],
                          "lds_over_lut": round(float(np.median(rows[15]) / np.median(rows[5])), 4),
                          "kernel_15": k15}), flush=True)
        del out
