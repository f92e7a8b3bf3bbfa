"""Does reading a window of input just before its decode pay?  Decoding the SAME window twice in a row
runs 11-16 % faster the second time at 2^13-2^15 frames (r06cp: the input comes out of the 256 MiB
memory-side cache; the 4 GB of nontemporal stores in between do not push it out).  Here: another
window at every launch; (a) the decode alone, (b) a sweep that merely reads the window (the header
search, nothing found) followed by the decode, on one stream; (c) the same with the sweep on a
second stream, the decode waiting for it -- as it would run next to the header scan of a read().
BB_EXPERIMENTS is not needed.    python tools/experiments/exp_touch_then_decode.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib, arena          # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (8 << 30) // bench.FRAME_NBYTES
image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
src = torch.arange(nframes, dtype=torch.int64, device=dev) * bench.FRAME_NBYTES + 32
ar = arena.enable()
side = torch.cuda.Stream()


def touch(first, nf):
    """Reads frames [first, first + nf) and keeps nothing: the Mark 5B sync search finds none in VDIF noise."""
    lo = first * bench.FRAME_NBYTES
    lo -= lo % 16
    kernels.mark5b_locate(image[lo:lo + nf * bench.FRAME_NBYTES], nf * bench.FRAME_NBYTES)


def run(mode, nf, out, r):
    nwin = max(1, (nframes - nf) // nf)
    first = ((r * 3 + 1) % nwin) * nf
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if mode == 'touch, same stream':
        touch(first, nf)
    elif mode == 'touch, side stream':
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            touch(first, nf)
        torch.cuda.current_stream().wait_stream(side)
    kernels.decode_frames(image, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=src[first:first + nf], out=out)
    b.record()
    b.synchronize()
    return a.elapsed_time(b)


for lg in (12, 13, 14, 15, 16, 17):
    nf = 1 << lg
    out = ar.empty((nf * bench.SPF,))
    row = []
    for mode in ('decode alone', 'touch, same stream', 'touch, side stream'):
        ts = [run(mode, nf, out, r) for r in range(9)][2:]
        row.append((mode, float(np.median(ts))))
    base = row[0][1]
    print("2^%d frames (%.2f GB out, %.0f MB in): " % (lg, nf * bench.SPF * 4 / 1e9, nf * bench.FRAME_NBYTES / 1e6)
          + "   ".join("%s %.1f us (%.3f of the peak%s)" % (m, ms * 1e3, nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / ms / 8e9,
                                                             "" if m == 'decode alone' else ", x%.3f" % (base / ms)) for m, ms in row), flush=True)
