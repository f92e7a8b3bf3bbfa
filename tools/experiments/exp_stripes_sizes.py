"""Which number of stripes (BB_TUNE_WORK_STRIPES) for which output size?  cfg2 decodes of
2^12 .. 2^19 frames (0.5 - 67 GB of output) into blocks of the product's arena and into plain
allocations, the launch dealt over 4, 8, 16, 32 stripes; medians over 4 outputs x 4 launches.
Needs the experiment build.    BB_EXPERIMENTS=1 python tools/experiments/exp_stripes_sizes.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GIB = 1 << 30
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib, placement       # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (8 << 30) // bench.FRAME_NBYTES
image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
ar = placement._arena_for(dev)


def rate(out, nf, first):
    ts = []
    for r in range(5):
        win = image[((first + r * 7919) % max(1, nframes - nf)) * bench.FRAME_NBYTES:][:nf * bench.FRAME_NBYTES]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32, src_stride=bench.FRAME_NBYTES, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / float(np.median(ts)) / 1e6


for lf in (12, 13, 14, 15, 16, 17, 18, 19):
    nf = 1 << lf
    n = nf * bench.SPF
    for what in ('arena', 'torch'):
        nout = 4 if lf <= 17 else 2
        outs = [ar.empty(n) if what == 'arena' else torch.empty(n, dtype=torch.float32, device=dev) for _ in range(nout)]
        row = []
        for lw in (2, 3, 4, 5):
            kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
            rr = [rate(o, nf, 31 * k) for k, o in enumerate(outs)]
            row.append((1 << lw, float(np.median(rr)), min(rr), max(rr)))
        del outs
        torch.cuda.empty_cache()
        print("2^%d frames (%.1f GB out) %-5s " % (lf, n * 4 / 1e9, what)
              + "  ".join("%2d: %.0f (%.0f-%.0f)" % r for r in row), flush=True)
