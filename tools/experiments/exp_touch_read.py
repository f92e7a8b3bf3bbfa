"""fh.read() of 2^12 .. 2^16 frames with and without the pre-read of the window (BB_TUNE_TOUCH_MIB:
256 = the product's default, 0 = never), within one process, alternating, another window of the file
image at every read: time until the read's samples are there (HIP events around the call).
Also Mark 5B.    python tools/experiments/exp_touch_read.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib, vdif           # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (8 << 30) // bench.FRAME_NBYTES
if os.environ.get('BB_EXP_IMAGE_IN_ARENA'):         # the file image in arena memory (mapped 32 MiB chunks), as fh.stage() keeps it
    from bench_legs.common import image_buffer
    image, where = image_buffer(nframes * bench.FRAME_NBYTES, dev)
    print('file image in', where, 'memory', flush=True)
else:
    image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
rate = bench.FRAME_RATE * bench.SPF
WALL = bool(os.environ.get('BB_EXP_WALL'))     # host wall clock between two device synchronisations instead of HIP events

with vdif.open(image, 'rs', sample_rate=rate) as fh:
    for lg in (13, 14, 15):
        nf = 1 << lg
        count = nf * bench.SPF
        nwin = nframes // nf - 1
        ts = {256: [], 0: []}
        for r in range(24 if lg < 12 else 16):
            for knob in (256, 0):
                kernels.tune(_lib.TUNE_TOUCH_MIB, knob)
                fh.seek(((r * 2 + (knob == 0) + 1) % nwin) * count)
                if WALL:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = fh.read(count)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) * 1e3
                else:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    out = fh.read(count)
                    b.record()
                    b.synchronize()
                    dt = a.elapsed_time(b)
                if r >= 3:
                    ts[knob].append(dt)
                del out
        on, off = float(np.median(ts[256])), float(np.median(ts[0]))
        nb = nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16)
        print("2^%d frames (%.2f GB out, %.0f MiB in): read with pre-read %.1f us (%.3f of the peak), without %.1f us (%.3f): x%.3f   [%s]"
              % (lg, count * 4 / 1e9, nf * bench.FRAME_NBYTES / 2 ** 20, on * 1e3, nb / on / 8e9, off * 1e3, nb / off / 8e9, off / on,
                 _lib.last_kernel()), flush=True)
kernels.tune(_lib.TUNE_TOUCH_MIB, -1)
