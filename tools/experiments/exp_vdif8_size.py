"""VDIF 8-bit (table levels) contiguous output: the plain kernel (product)
against k_decode_flat_lds<8,LDS,glds> with 16 / 8 tiles per wave, BY INPUT SIZE
(round 4 saw -6 % at 8 GiB and +3 % at 31 GiB and wrote no
size switch).  Same process, same buffers, bit-identity checked; the output is
an arena block up to 64 GiB, a plain tensor above.
    BB_EXPERIMENTS=1 python tools/experiments/exp_vdif8_size.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402
import baseband_amd                             # noqa: E402

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


def ms_of(fn, reps=5):
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


def setk(variant=5, tiles=0, flat8=0):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    kernels.tune(_lib.TUNE_LUT_TILES, tiles)
    kernels.tune(_lib.TUNE_FLAT8_LDS, flat8)


frame, pay, hdr = 8032, 8000, 32
for coder, cname in ((_lib.CODER_VDIF, "VDIF 8-bit (table)"), (_lib.CODER_INT, "int8 frames of 8000 B")):
    for gib in (0.5, 1, 2, 4, 8, 12, 16, 24, 31):
        nfr = int(gib * 2 ** 30) // frame
        n = nfr * pay
        o = baseband_amd.empty_output((n,), dtype=torch.float32, device=dev)
        alg = nfr * (frame + pay * 4)
        res, kn, ref = {}, {}, None
        same = True
        arms = [("plain", dict(flat8=2)), ("glds16", dict(flat8=1, variant=20, tiles=4)),
                ("glds8", dict(flat8=1, variant=20, tiles=2)), ("glds4", dict(flat8=1, variant=20, tiles=1))]
        for rnd in range(3):
            for label, kw in arms:
                setk(**kw)
                ms = ms_of(lambda: kernels.decode_frames(buf, nfr, pay, coder, 8, src0=hdr, src_stride=frame, out=o))
                res.setdefault(label, []).append(round(alg / ms / 1e6 / 8000, 4))
                kn[label] = _lib.last_kernel().split(' grid')[0]
                if rnd == 0:
                    d = int(o.view(torch.int32)[::977].to(torch.int64).sum().item())
                    ref = d if ref is None else ref
                    same &= d == ref
        setk()
        med = {k: float(np.median(r)) for k, r in res.items()}
        print(json.dumps({"case": cname, "GiB_in": gib, "frac": med, "over_plain": {k: round(m / med["plain"], 4) for k, m in med.items()},
                          "identical": same, "kernels": kn}), flush=True)
        del o
