"""The pre-read of the window (exp_touch_read.py) for the 8-thread layouts: 8 threads x 1 channel real
(sample.vdif's) and 8 threads x 16 channels complex (cfg3), fh.read() of 2^10 / 2^11 / 2^12 frame sets
(2^13-2^15 frames), alternating with and without, another window at every read."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib, vdif           # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nthread = 8
nsets = (4 << 30) // (bench.FRAME_NBYTES * nthread)
for nchan, cplx, name in ((1, False, '8 threads x 1 channel, real'), (16, True, '8 threads x 16 channels, complex')):
    image, h0 = bench.make_file_image_on_device(nsets, 7, 0, dev, nthread=nthread, nchan=nchan, complex_data=cplx,
                                                order=tuple(range(nthread)))
    spf = bench.PAYLOAD_NBYTES * 8 // 2 // nchan // (2 if cplx else 1)
    rate = bench.FRAME_RATE * spf
    with vdif.open(image, 'rs', sample_rate=rate) as fh:
        for lg in (10, 11, 12):
            ns = 1 << lg
            count = ns * spf
            nwin = nsets // ns - 1
            ts = {256: [], 0: []}
            for r in range(14):
                for knob in (256, 0):
                    kernels.tune(_lib.TUNE_TOUCH_MIB, knob)
                    fh.seek(((r * 2 + (knob == 0) + 1) % nwin) * count)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    out = fh.read(count)
                    b.record()
                    b.synchronize()
                    if r >= 3:
                        ts[knob].append(a.elapsed_time(b))
                    del out
            on, off = float(np.median(ts[256])), float(np.median(ts[0]))
            print("%s, 2^%d sets (%.0f MiB in): with pre-read %.1f us, without %.1f us: x%.3f   [%s]"
                  % (name, lg, ns * nthread * bench.FRAME_NBYTES / 2 ** 20, on * 1e3, off * 1e3, off / on, _lib.last_kernel()), flush=True)
    del image
kernels.tune(_lib.TUNE_TOUCH_MIB, -1)
