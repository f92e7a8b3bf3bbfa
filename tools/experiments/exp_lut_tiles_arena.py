"""Tiles per wave of the headline kernel (BB_TUNE_LUT_TILES) for mid-size launches
whose output and input lie in the arena; next window of the image per launch.
    python tools/experiments/exp_lut_tiles_arena.py
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
nxt = [0]


def rate(out, nf):
    ts = []
    for r in range(5):
        f0 = nxt[0]
        nxt[0] = (nxt[0] + nf) % (nframes - nf)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(image, nf, 8000, _lib.CODER_VDIF, 2, src0=32 + f0 * 8032, src_stride=8032, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return round(nf * (8032 + 128000) / float(np.median(ts)) / 1e9, 3)


for nf in (1 << 15, 1 << 16, 1 << 18):
    res = {}
    held = []
    for draw in range(3):
        out = ar.empty(nf * 32000)
        for tiles in (4, 2, 3, 5, 6, 8, 4):
            kernels.tune(_lib.TUNE_LUT_TILES, tiles)
            res.setdefault(str(tiles), []).append(rate(out, nf))
        held.append(ar.empty((64 << 20) // 4))
        del out
    del held
    kernels.tune(_lib.TUNE_LUT_TILES, 4)
    print(json.dumps({"frames": nf, "TBps_by_tiles_per_wave": res}), flush=True)
