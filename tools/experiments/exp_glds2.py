"""(Knob values as of the round-4 promotion: BB_TUNE_LUT_TILES 0 = the default = 6 tiles per wave
for the 2-bit kernel; BB_TUNE_FLAT8_LDS 2 = the plain 8-bit kernel.)
Round 4: direct-to-LDS loads (global_load_lds_dwordx4) beyond the headline
case.  Same process, same buffers, interleaved repeats, bit-identity checked:
  (1) 2-bit headline launch: tiles per wave 2 / 4 / 6 / 8 with the glds kernel;
  (2) 1- and 4-bit contiguous output: k_decode_flat_lut (product) against
      k_decode_flat_lds with register staging (variant 15) and with glds (20);
  (3) 8-bit contiguous output (DADA int8, VDIF 8-bit): the plain kernel
      (product) against k_decode_flat_lds<8> staged (FLAT8_LDS, variant 5) and
      with glds (variant 20), 16 / 8 tiles per wave.
    BB_EXPERIMENTS=1 python tools/experiments/exp_glds2.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402

assert _lib.EXPERIMENTS, "run with BB_EXPERIMENTS=1"
dev = torch.device('cuda', 0)
kernels.init()
big = 31 << 30
buf = torch.empty(big + 4096, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(3)
v = buf[:big].view(torch.int32)
for lo in range(0, v.numel(), 1 << 28):
    hi = min(v.numel(), lo + (1 << 28))
    v[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
out = torch.empty(34_222_816_000, dtype=torch.float32, device=dev)


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


def compare(name, frame, pay, hdr, coder, bps, limit, arms, rounds=3):
    nfr = min(int(limit) // frame, out.numel() // (pay * 8 // bps))
    o = out[:nfr * (pay * 8 // bps)]
    alg = nfr * (frame + pay * 8 // bps * 4)
    res, kn, dg = {}, {}, {}
    for rnd in range(rounds):
        for label, kw in arms:
            setk(**kw)
            ms = ms_of(lambda: kernels.decode_frames(buf, nfr, pay, coder, bps, src0=hdr, src_stride=frame, out=o))
            res.setdefault(label, []).append(round(alg / ms / 1e6 / 8000, 4))
            kn[label] = _lib.last_kernel().split(' grid')[0]
            if rnd == 0:
                dg[label] = digest(o)
    setk()
    first = arms[0][0]
    print(json.dumps({"case": name, "frames": nfr, "frac_of_8TBps": res, "kernels": kn,
                      "bit_identical": all(d == dg[first] for d in dg.values()),
                      "median_over_" + first: {k: round(float(np.median(r) / np.median(res[first])), 4) for k, r in res.items()}}),
          flush=True)


G = 2 ** 30
compare("VDIF 2-bit 8032-byte frames (headline shape), glds kernel, tiles per wave", 8032, 8000, 32, _lib.CODER_VDIF, 2, 8 * G,
        [("tiles4", dict(tiles=4)), ("tiles2", dict(tiles=2)), ("tiles6", dict(tiles=6)), ("tiles8", dict(tiles=8)),
         ("regs_tiles4", dict(variant=19, tiles=4))])
compare("Mark 5B 2-bit 10016-byte frames", 10016, 10000, 16, _lib.CODER_MARK5B, 2, 8 * G,
        [("glds", dict()), ("regs", dict(variant=19))])
for bps, lim in ((1, 4 * G), (4, 16 * G)):
    compare("VDIF {}-bit 8032-byte frames".format(bps), 8032, 8000, 32, _lib.CODER_VDIF, bps, lim,
            [("lut_product", dict()), ("lds_regs", dict(variant=15)), ("lds_glds", dict(variant=20))])
for name, coder, frame, pay, hdr in (("DADA int8, 128 MiB frames", _lib.CODER_INT, 128 << 20, (128 << 20) - 4096, 4096),
                                     ("VDIF 8-bit, 8032-byte frames", _lib.CODER_VDIF, 8032, 8000, 32)):
    compare(name, frame, pay, hdr, coder, 8, big,
            [("plain", dict(flat8=2)), ("lds_regs_16", dict(flat8=1, tiles=4)), ("lds_glds_16", dict(variant=20, flat8=1, tiles=4)),
             ("lds_glds_8", dict(variant=20, flat8=1, tiles=2)), ("lds_glds_4", dict(variant=20, flat8=1, tiles=1))])
