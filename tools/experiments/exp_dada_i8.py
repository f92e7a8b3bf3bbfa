"""DADA int8 (flat int8 -> float32): the product's k_decode_flat_lds<8,INT8,glds> with 4
tiles per wave against the plain kernel and 8 / 16 tiles, at the bench's shape (ONE run
of 31 GiB), as 128 MiB frames behind 4096-byte headers, and at 8 GiB (VERDICT r4 next
4c: 0.81 -> 0.83 asked).  Same process, same buffers, bit-identity checked.
    BB_EXPERIMENTS=1 python tools/experiments/exp_dada_i8.py"""
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
buf = torch.empty(big + (1 << 20), dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(3)
v = buf[:big].view(torch.int32)
for lo in range(0, v.numel(), 1 << 28):
    hi = min(v.numel(), lo + (1 << 28))
    v[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
out = torch.empty(big, dtype=torch.float32, device=dev)


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


for name, nfr, pay, hdr in (("one run of 31 GiB (bench row)", 1, big, 0), ("128 MiB frames, 4096-byte headers, 31 GiB", 247, 128 << 20, 4096),
                            ("one run of 8 GiB", 1, 8 << 30, 0), ("128 MiB frames, 8 GiB", 63, 128 << 20, 4096)):
    o = out[:nfr * pay]
    alg = nfr * (pay + hdr) + nfr * pay * 4
    res, kn, ref, same = {}, {}, None, True
    for rnd in range(3):
        for label, kw in (("product_glds4", dict()), ("plain", dict(flat8=2)), ("glds8", dict(flat8=1, variant=20, tiles=2)),
                          ("glds16", dict(flat8=1, variant=20, tiles=4)), ("glds2", dict(flat8=1, variant=20, tiles=0))):
            setk(**kw)
            if label == "glds2":
                kernels.tune(_lib.TUNE_LUT_TILES, 0)
                continue
            ms = ms_of(lambda: kernels.decode_frames(buf, nfr, pay, _lib.CODER_INT, 8, src0=hdr, src_stride=pay + hdr, out=o))
            res.setdefault(label, []).append(round(alg / ms / 1e6 / 8000, 4))
            kn[label] = _lib.last_kernel()
            if rnd == 0:
                d = int(o.view(torch.int32)[::977].to(torch.int64).sum().item())
                ref = d if ref is None else ref
                same &= d == ref
    setk()
    med = {k: float(np.median(r)) for k, r in res.items()}
    print(json.dumps({"case": name, "frac": med, "over_product": {k: round(m / med["product_glds4"], 4) for k, m in med.items()},
                      "identical": same, "kernels": kn}), flush=True)
