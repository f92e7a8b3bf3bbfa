"""Tiles per wave of the 2-bit headline kernel by OUTPUT SIZE and by where the output lies (an arena
block, a fresh torch.empty): round 6 found 2 tiles (16 KiB of output per work item) 9-13 % ahead of
the product's 6 at 2^15 frames into a slice of a large plain allocation, and far behind at the
headline size (exp_tiles_under_8_stripes.py).  Three fresh outputs per cell, median of the three
medians; within one process.
Needs the experiment build.    BB_EXPERIMENTS=1 python tools/experiments/exp_tiles_by_size.py"""
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


def timed(fn, reps=5):
    """`fn(r)` decodes ANOTHER window of the image every time: a window decoded twice in a row is
    served by the 256 MiB memory-side cache the second time, which flatters small work items."""
    ts = []
    for r in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(r)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


TILES = (1, 2, 3, 4, 6)
for lg in (13, 14, 15, 16, 17, 18):
    nf = 1 << lg
    for kind in ('arena', 'torch'):
        cells = {t: [] for t in TILES}
        keep = []
        for draw in range(3):
            o = ar.empty((nf * bench.SPF,)) if kind == 'arena' else torch.empty(nf * bench.SPF, dtype=torch.float32, device=dev)
            keep.append(o)                                  # (hold them: the next draw lies elsewhere)
            nwin = max(1, (nframes - nf) // nf)
            for t in TILES:
                kernels.tune(_lib.TUNE_LUT_TILES, t)

                def one(r, t=t):
                    first = ((draw * 7 + t * 5 + r) % nwin) * nf
                    kernels.decode_frames(image, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=src[first:first + nf], out=o)
                ms = timed(one)
                cells[t].append(nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / ms / 8e9)
        del keep
        print("2^%d frames (%5.2f GB out) %-5s  " % (lg, nf * bench.SPF * 4 / 1e9, kind)
              + "   ".join("%d: %.3f [%.3f-%.3f]" % (t, float(np.median(v)), min(v), max(v)) for t, v in cells.items()), flush=True)
kernels.tune(_lib.TUNE_LUT_TILES, 0)
