"""Reading only a PREFIX of the window through before its decode: does a window near or above the
cache's 256 MiB gain from having its first 160 / 192 / 224 MiB there?  fh.read() of 2^15 (251 MiB) and
2^16 frames (502 MiB), one process, another window at every read.  (Knob 44 exists only while this
experiment does.)"""
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
nframes = (8 << 30) // bench.FRAME_NBYTES
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev)
rate = bench.FRAME_RATE * bench.SPF
CASES = (('no pre-read', 0, 0), ('whole window', 1024, 0), ('first 224 MiB', 1024, 224), ('first 192 MiB', 1024, 192),
         ('first 160 MiB', 1024, 160), ('first 128 MiB', 1024, 128))
with vdif.open(image, 'rs', sample_rate=rate) as fh:
    for lg in (15, 16):
        nf = 1 << lg
        count = nf * bench.SPF
        nwin = nframes // nf - 1
        ts = {c[0]: [] for c in CASES}
        for r in range(13):
            for i, (name, lim, pre) in enumerate(CASES):
                kernels.tune(_lib.TUNE_TOUCH_MIB, lim)
                kernels.tune(44, pre)
                fh.seek(((r * len(CASES) + i + 1) % nwin) * count)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = fh.read(count)
                b.record()
                b.synchronize()
                if r >= 3:
                    ts[name].append(a.elapsed_time(b))
                del out
        base = float(np.median(ts['no pre-read']))
        print("2^%d frames (%d MiB in): " % (lg, nf * bench.FRAME_NBYTES >> 20)
              + "   ".join("%s %.1f us (x%.3f)" % (k, float(np.median(v)) * 1e3, base / float(np.median(v))) for k, v in ts.items()), flush=True)
kernels.tune(_lib.TUNE_TOUCH_MIB, -1)
kernels.tune(44, 0)
