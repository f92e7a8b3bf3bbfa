"""Stripes (BB_TUNE_WORK_STRIPES) for the families round 6 left on the old rule: channel
selections (k_decode_pick / gather_select), Mark 4, the frame copy -- the same launch, same
buffers, under 4 / 8 / 16 stripes, at a small and a large input.
Needs the experiment build.    BB_EXPERIMENTS=1 python tools/experiments/exp_stripes_others.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib, arena           # noqa: E402
from baseband_amd.mark4._bitmaps import BITMAPS         # noqa: E402

kernels.init()
dev = torch.device('cuda')
ar = arena.Arena(160 << 30)
out = ar.empty(34 << 30)                   # 136 GB of float32
buf = torch.randint(0, 256, ((8 << 30) + 4096,), dtype=torch.uint8, device=dev)


def timed(fn, reps=5):
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


def sweep(name, fn, nbytes_moved):
    row = []
    for lw in (-1, 2, 3, 4):
        kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
        ms = timed(fn)
        row.append("%s: %.0f" % ('product' if lw < 0 else str(1 << lw), nbytes_moved / ms / 1e6))
    kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
    print("%-64s GB/s  %s   [%s]" % (name, "  ".join(row), _lib.last_kernel().split(' grid')[0]), flush=True)


for gib in (0.5, 8):
    nbytes = int(gib * (1 << 30))
    # 2 of 16 channels of 8 threads x 16 channels complex (k_decode_pick)
    fn_, pn, nth = 8032, 8000, 8
    nsets = nbytes // (fn_ * nth)
    src8 = torch.arange(nsets * nth, device=dev, dtype=torch.int64) * fn_ + 32
    w = torch.tensor([6, 7, 8, 9], dtype=torch.int32, device=dev)
    n = nsets * 1000 * nth * 4
    sweep("%.1f GiB in: VDIF 8 thr x 16 ch complex, 2 of 16 channels" % gib,
          lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src8, complex_data=True, out=out[:n], within=w),
          nsets * nth * fn_ + n * 4)
    # Mark 4, 64 tracks
    maps = BITMAPS[(8, 2, 4)]
    nf4 = nbytes // 160000
    sweep("%.1f GiB in: Mark 4 64 tracks fanout 4" % gib,
          lambda: kernels.decode_mark4(buf, nf4, 64, 20000, maps['sign_bit'], maps['mag_bit'], fill_words=160, src0=0,
                                       src_stride=160000, out=out[:nf4 * 640000]),
          nf4 * 160000 + nf4 * 640000 * 4)
    # frame copy (DADA NBIT 32)
    fb = 4096 + (64 << 20)
    nfc = nbytes // fb
    if nfc:
        o8 = out.view(torch.uint8)
        sweep("%.1f GiB in: frame copy, 64 MiB payloads" % gib,
              lambda: kernels.copy_frames(buf, nfc, 64 << 20, src0=4096, src_stride=fb, out=o8[:nfc * (64 << 20)]),
              2 * nfc * (64 << 20))
    # Mark 5B 16 channels, 2 of them (gather_select / pick on one slot)
    nfr = nbytes // 10016
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * 10016 + 16
    w2 = torch.tensor([1, 6], dtype=torch.int32, device=dev)
    n2 = nfr * 2500 * 2
    sweep("%.1f GiB in: Mark 5B 16 ch, 2 of 16 channels" % gib,
          lambda: kernels.decode_frames(buf, nfr, 10000, _lib.CODER_MARK5B, 2, chunk=16, src=src, out=out[:n2], within=w2),
          nfr * 10016 + n2 * 4)
