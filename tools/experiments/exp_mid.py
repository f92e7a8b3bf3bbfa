"""Round 4: the launch sizes of ordinary read() calls (2^15 .. 2^18 cfg2 frames,
fresh arena blocks, every launch on the next window of the 8 GiB image) with
the round-4 kernel choices: direct-to-LDS loads with 6 (default) and 4 tiles
per wave against round 3's register-staged form with 4.  Arms interleaved per
block; per arm the median of 6 launches; 4 blocks per size.
    BB_EXPERIMENTS=1 python tools/experiments/exp_mid.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib, arena           # noqa: E402

assert _lib.EXPERIMENTS, "run with BB_EXPERIMENTS=1"
dev = torch.device('cuda', 0)
kernels.init()
FRAME, PAY, HDR = 8032, 8000, 32
nframes = (8 << 30) // FRAME
ar = arena.Arena(250 << 30)
image = ar.empty(nframes * FRAME, dtype=torch.uint8)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
nxt = [0]


def rate(out, nf, variant, tiles, launches=6):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    kernels.tune(_lib.TUNE_LUT_TILES, tiles)
    ts = []
    for r in range(launches + 1):
        if nxt[0] + nf > nframes:
            nxt[0] = 0
        first = nxt[0]
        nxt[0] += nf
        win = image[first * FRAME:(first + nf) * FRAME]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, PAY, _lib.CODER_VDIF, 2, src0=HDR, src_stride=FRAME, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
    kernels.tune(_lib.TUNE_LUT_TILES, 0)
    return round(nf * (FRAME + PAY * 16) / float(np.median(ts)) / 1e6 / 8000, 4)


arms = (("glds_t6", 5, 0), ("glds_t4", 5, 4), ("glds_t8", 5, 8), ("regs_t4", 19, 4))
print(json.dumps({"arena": {k: ar.stats()[k] for k in ("bytes_backed", "probes", "last_probe_gbps", "grow_ms")}}), flush=True)
for lf in (14, 15, 16, 18):
    nf = 1 << lf
    res = {a[0]: [] for a in arms}
    held = []
    for blk in range(4):
        o = ar.empty(nf * 32000)
        for name, v, t in arms:
            res[name].append(rate(o, nf, v, t))
        held.append(ar.empty((64 << 20) // 4))
        del o
    del held
    print(json.dumps({"frames": nf, "output_GB": round(nf * 128000 / 1e9, 2), "frac_of_8TBps_per_block": res,
                      "median": {k: float(np.median(v)) for k, v in res.items()}}), flush=True)
print(json.dumps({"arena_after": {k: ar.stats()[k] for k in ("bytes_backed", "probes", "last_probe_gbps", "grow_ms", "steps")}}), flush=True)
