#!/usr/bin/env python3
"""Launch-size cliff, round 2: the explicit write front (k_decode_flat_front,
variants 6-9) against the persistent kernel (5) and the plain one (0), same
process, same buffers, launches of 2^16 .. 2^20 frames of the cfg2 layout --
on a slice of one big output allocation and on a fresh exactly-sized one."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
payload, header = 8000, 32
stride = payload + header
nmax = 1 << 20
quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
buf = torch.randint(0, 256, (nmax * stride + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nmax * payload * 4, dtype=torch.float32, device='cuda')
per = payload * 4


def setv(v, G=2048, K=16):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, v)
    kernels.tune(_lib.TUNE_FRONT_GROUP, G)
    kernels.tune(_lib.TUNE_FRONT_STEPS, K)


def run(nfr, o, v, G=2048, K=16):
    setv(v, G, K)
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                              src_stride=stride, out=o), reps=5)
    return round(nfr * (stride + payload * 16) / ms / 1e9, 3)


# correctness of every variant against the plain kernel, 2^14 frames
setv(0)
ref = kernels.decode_frames(buf, 1 << 14, payload, _lib.CODER_VDIF, 2, src0=header, src_stride=stride).clone()
for v in (6, 7, 8, 9):
    for G, K in ((2048, 16), (1000, 7)):
        setv(v, G, K)
        got = kernels.decode_frames(buf, 1 << 14, payload, _lib.CODER_VDIF, 2, src0=header, src_stride=stride)
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (v, G, K)
print(json.dumps({"front variants bit-identical to the plain kernel": True}), flush=True)

sizes = (1 << 16, 1 << 18, 1 << 20) if quick else (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20)
grid = [(G, K) for G in (1024, 2048, 4096) for K in (4, 16, 64)]
for nfr in sizes:
    o = out[:nfr * per]
    row = {"frames": nfr, "where": "first part of the 134 GB output", "v5_persistent": run(nfr, o, 5),
           "v0_plain": run(nfr, o, 0)}
    for v in (6, 7, 8, 9):
        for G, K in grid:
            row["v%d_G%d_K%d" % (v, G, K)] = run(nfr, o, v, G, K)
    print(json.dumps(row), flush=True)
# more extreme fronts for the best-looking variant family
for nfr in (1 << 18, 1 << 20):
    o = out[:nfr * per]
    row = {"frames": nfr, "where": "first part, wide sweep"}
    for v in (6, 7):
        for G, K in ((256, 64), (512, 256), (2048, 256), (8192, 4), (8192, 16), (16384, 8), (65536, 4)):
            row["v%d_G%d_K%d" % (v, G, K)] = run(nfr, o, v, G, K)
    print(json.dumps(row), flush=True)
del out
torch.cuda.empty_cache()
for nfr in sizes:
    o = torch.empty(nfr * per, dtype=torch.float32, device='cuda')
    row = {"frames": nfr, "where": "fresh exactly-sized allocation", "v5_persistent": run(nfr, o, 5),
           "v0_plain": run(nfr, o, 0)}
    for v in (6, 7, 8, 9):
        for G, K in ((2048, 16), (4096, 16), (2048, 64)):
            row["v%d_G%d_K%d" % (v, G, K)] = run(nfr, o, v, G, K)
    ms = timeit(lambda: o.fill_(1.0), reps=5)
    row["torch_fill"] = round(o.numel() * 4 / ms / 1e9, 3)
    print(json.dumps(row), flush=True)
    del o
    torch.cuda.empty_cache()
setv(5)
