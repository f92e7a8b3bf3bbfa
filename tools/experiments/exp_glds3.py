"""(Knob values as of the round-4 promotion: BB_TUNE_LUT_TILES 0 = the default = 6 tiles per wave
for the 2-bit kernel; BB_TUNE_FLAT8_LDS 2 = the plain 8-bit kernel.)
Round 4: is the direct-to-LDS form worth keeping?  A/Bs of +-1.5 % flip
between boxes and allocations, so this script is run on SEVERAL fresh boxes
(profiles/r04d_exp_glds3_box*.log) under the conditions of bench.py: the 8 GiB
cfg2 image in arena memory, its index, the 127.5 GiB torch.empty output;
arms interleaved, 4 rounds, median of 4 launches each, digests compared.
    BB_EXPERIMENTS=1 python tools/experiments/exp_glds3.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib                  # noqa: E402

assert _lib.EXPERIMENTS, "run with BB_EXPERIMENTS=1"
dev = torch.device('cuda', 0)
kernels.init()
FRAME, PAY, HDR = 8032, 8000, 32
nframes = (8 << 30) // FRAME
image, where = bench.image_buffer(nframes * FRAME, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
out = bench.empty_with_patience(nframes * 32000, torch.float32, dev)
src = torch.arange(nframes, device=dev, dtype=torch.int64) * FRAME + HDR


def ms_of(fn, reps=4):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def digest(o, n=1 << 26):
    w = o.view(torch.int32)
    m = w.numel()
    return [int(w[k:k + n].to(torch.int64).sum().item()) for k in (0, (m // 2) & ~3, m - n)]


def setk(variant=5, tiles=0, flat8=0):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    kernels.tune(_lib.TUNE_LUT_TILES, tiles)
    kernels.tune(_lib.TUNE_FLAT8_LDS, flat8)


def compare(name, fn, alg, arms, rounds=4):
    res, kn, dg = {}, {}, {}
    for rnd in range(rounds):
        for label, kw in arms:
            setk(**kw)
            ms = ms_of(fn)
            res.setdefault(label, []).append(round(alg / ms / 1e6 / 8000, 4))
            kn[label] = _lib.last_kernel().split(' grid')[0]
            if rnd == 0:
                dg[label] = digest(out)
    setk()
    first = arms[0][0]
    print(json.dumps({"case": name, "image_memory": where, "frac_of_8TBps": res, "kernels": kn,
                      "bit_identical": all(d == dg[first] for d in dg.values()),
                      "median_over_" + first: {k: round(float(np.median(r) / np.median(res[first])), 4) for k, r in res.items()}}),
          flush=True)


compare("headline: cfg2 image + index -> 127.5 GiB",
        lambda: kernels.decode_frames(image, nframes, PAY, _lib.CODER_VDIF, 2, src=src, out=out),
        nframes * (FRAME + PAY * 16),
        [("regs_t4", dict(variant=19, tiles=4)), ("glds_t4", dict(tiles=4)), ("regs_t6", dict(variant=19, tiles=6)), ("glds_t6", dict())])
compare("the same without the index (fixed stride)",
        lambda: kernels.decode_frames(image, nframes, PAY, _lib.CODER_VDIF, 2, src0=HDR, src_stride=FRAME, out=out),
        nframes * (FRAME + PAY * 16),
        [("regs_t4", dict(variant=19, tiles=4)), ("glds_t4", dict(tiles=4)), ("regs_t6", dict(variant=19, tiles=6)), ("glds_t6", dict())])
o4 = out[:nframes * 16000]
compare("VDIF 4-bit, the same image",
        lambda: kernels.decode_frames(image, nframes, PAY, _lib.CODER_VDIF, 4, src0=HDR, src_stride=FRAME, out=o4),
        nframes * (FRAME + PAY * 8),
        [("lut_product", dict()), ("lds_glds", dict(variant=20)), ("lds_glds_t6", dict(variant=20, tiles=3))])
o8 = out[:nframes * 8000]
compare("VDIF 8-bit, the same image",
        lambda: kernels.decode_frames(image, nframes, PAY, _lib.CODER_VDIF, 8, src0=HDR, src_stride=FRAME, out=o8),
        nframes * (FRAME + PAY * 4),
        [("plain", dict(flat8=2)), ("lds_glds_16", dict(variant=20, flat8=1, tiles=4)), ("lds_glds_8", dict(variant=20, flat8=1, tiles=2))])
nb = (image.numel() - 4096) // 16 * 16
oi = out[:nb]
compare("DADA int8: the image as one 8 GiB payload",
        lambda: kernels.decode_frames(image, 1, nb, _lib.CODER_INT, 8, src0=4096, out=oi),
        nb * 5,
        [("plain", dict(flat8=2)), ("lds_glds_16", dict(variant=20, flat8=1, tiles=4)), ("lds_glds_8", dict(variant=20, flat8=1, tiles=2)),
         ("lds_glds_4", dict(variant=20, flat8=1, tiles=1))])
