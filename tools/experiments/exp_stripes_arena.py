"""Work order of mid-size launches whose output lies in the ARENA: number of
stripes the work items are dealt over (BB_TUNE_WORK_STRIPES: log2; default 4
stripes below 16 GiB of output, 16 above), cfg2 frames, fresh block per draw.
    python tools/experiments/exp_stripes_arena.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402
import bench                                            # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (8 << 30) // bench.FRAME_NBYTES
image, _ = bench.image_buffer(nframes * bench.FRAME_NBYTES, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
ar = arena.default()


def rate(out, nf, f0):
    ts = []
    for r in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(image, nf, 8000, _lib.CODER_VDIF, 2, src0=32 + (f0 + r * 1000) % (nframes - nf) * 8032,
                              src_stride=8032, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return round(nf * (8032 + 128000) / float(np.median(ts)) / 1e9, 3)


for nf in (1 << 15, 1 << 16, 1 << 18):
    res = {}
    held = []
    for draw in range(4):
        out = ar.empty(nf * 32000)
        for lw in (-1, 0, 1, 2, 3, 4, 5, 6, 8):
            kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
            res.setdefault("default" if lw < 0 else str(1 << lw), []).append(rate(out, nf, draw * 150001))
        held.append(ar.empty((64 << 20) // 4))
        del out
    del held
    kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
    print(json.dumps({"frames": nf, "TBps_by_stripes": res}), flush=True)
