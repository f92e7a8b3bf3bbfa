"""Tiles per wave of the headline kernel again, now that launches are dealt over 8 stripes (round 4
chose 6 tiles = 48 KiB of output per work item under 16 stripes; a kernel that only writes prefers
16 KiB items under 8 stripes: tools/experiments/write_probe.cpp).  Within one process, same image,
same output, through an index like the drop-in path.
Needs the experiment build.    BB_EXPERIMENTS=1 python tools/experiments/exp_tiles_under_8_stripes.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib                  # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (8 << 30) // bench.FRAME_NBYTES
image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
out = torch.empty(nframes * bench.SPF, dtype=torch.float32, device=dev)
src = torch.arange(nframes, dtype=torch.int64, device=dev) * bench.FRAME_NBYTES + 32


def timed(fn, reps=6):
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


nbytes = nframes * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16)
for round_ in range(2):
    for nf, label in ((nframes, 'headline'), (1 << 15, '2^15 frames')):
        row = []
        for tiles in (2, 3, 4, 6, 8):
            kernels.tune(_lib.TUNE_LUT_TILES, tiles)
            o = out[:nf * bench.SPF]
            s = src[:nf]
            ms = timed(lambda: kernels.decode_frames(image, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=s, out=o))
            row.append("%d: %.4f (%s)" % (tiles, nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / ms / 8e9,
                                          _lib.last_kernel().split('tiles/wave')[-1].strip()))
        print("%-12s fraction of 8 TB/s by tiles per wave   %s" % (label, "   ".join(row)), flush=True)
kernels.tune(_lib.TUNE_LUT_TILES, 0)
