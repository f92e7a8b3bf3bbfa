"""8, 16 or 32 stripes at the headline size and at 34 / 67 GB of output (the product: 16 from
16 GiB of output on)?  The cfg2 image of 8 GiB decoded whole into one 127.5 GiB tensor, and
parts of it into 2^18 / 2^19-frame outputs; also the 8-thread cfg3 layout and Mark 5B at 8 GiB in.
Needs the experiment build.    BB_EXPERIMENTS=1 python tools/experiments/exp_stripes_headline.py"""
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


for nf in (nframes, 1 << 19, 1 << 18):
    row = []
    for lw in (3, 4, 5, 6):
        kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
        ms = timed(lambda: kernels.decode_frames(image, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32,
                                                 src_stride=bench.FRAME_NBYTES, out=out[:nf * bench.SPF]))
        row.append("%2d: %.0f" % (1 << lw, nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / ms / 1e6))
    print("cfg2 %8d frames (%.1f GB out): GB/s by stripes  %s" % (nf, nf * bench.SPF * 4 / 1e9, "  ".join(row)), flush=True)
# the same launch through an index (the drop-in path's form)
src = torch.arange(nframes, dtype=torch.int64, device=dev) * bench.FRAME_NBYTES + 32
row = []
for lw in (3, 4, 5):
    kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
    ms = timed(lambda: kernels.decode_frames(image, nframes, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=src, out=out))
    row.append("%2d: %.0f" % (1 << lw, nframes * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / ms / 1e6))
print("cfg2 headline through an index: GB/s by stripes  %s" % "  ".join(row), flush=True)
# 8 threads x 16 channels complex (cfg3), the gather kernel
nsets = nframes // 8
src8 = (torch.arange(nsets * 8, dtype=torch.int64, device=dev) * bench.FRAME_NBYTES + 32)
row = []
for lw in (3, 4, 5):
    kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
    ms = timed(lambda: kernels.decode_frames(image, nsets, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, chunk=32, nslot=8,
                                             src=src8, complex_data=True, out=out[:nsets * 8 * bench.SPF]))
    row.append("%2d: %.0f" % (1 << lw, nsets * 8 * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / ms / 1e6))
print("cfg3 layout (8 slots x 32 floats), 8 GiB in: GB/s by stripes  %s" % "  ".join(row), flush=True)
kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
