#!/usr/bin/env python3
"""Follow-up to exp_stripe.py: WHICH distance between concurrently written
regions lifts a launch from 5.6 to 6.5 TB/s?  2^16-frame launches (8.4 GB of
output) dealt over W stripes that lie D apart, D swept from 0.5 GiB to 64 GiB;
plus: does it matter where in the 134 GB buffer the stripes start?"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
payload, header = 8000, 32
stride = payload + header
nmax = 1 << 20
buf = torch.randint(0, 256, (nmax * stride + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nmax * payload * 4, dtype=torch.float32, device='cuda')
per = payload * 4
print(json.dumps({"out_ptr": hex(out.data_ptr()), "buf_ptr": hex(buf.data_ptr())}), flush=True)


def run(nfr, o, W, S, variant=5):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    kernels.tune(_lib.TUNE_OUT_STRIPE_W, W)
    kernels.tune(_lib.TUNE_OUT_STRIPE_S, S)
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                              src_stride=stride, out=o), reps=5)
    kernels.tune(_lib.TUNE_OUT_STRIPE_W, 0)
    return round(nfr * (stride + payload * 16) / ms / 1e9, 3)


nfr = 1 << 16
GiB = 2 ** 30
slot = per * 4                      # bytes per frame slot (128000)
for W in (2, 3, 4, 8):
    row = {"frames": nfr, "W": W}
    for d_gib in (0.5, 1, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 64):
        S = int(d_gib * GiB) // slot
        if (W - 1) * S + nfr // W + 1 > nmax:
            continue
        row["D=%gGiB" % d_gib] = run(nfr, out, W, S)
    print(json.dumps(row), flush=True)
# where the pair of stripes starts: W=2, D = 16 and 32 GiB, base offset swept
for d_gib in (16, 32):
    S = int(d_gib * GiB) // slot
    row = {"frames": nfr, "W": 2, "D_GiB": d_gib}
    for base_gib in (0, 4, 8, 12, 16, 24, 32, 48, 64, 80):
        b = int(base_gib * GiB) // slot
        if b + S + nfr // 2 + 1 > nmax:
            continue
        row["base=%dGiB" % base_gib] = run(nfr, out[b * per:], 2, S)
    print(json.dumps(row), flush=True)
# a narrow contiguous launch at different bases (is any single region fast by itself?)
row = {"frames": nfr, "contiguous": True}
for base_gib in range(0, 120, 8):
    b = int(base_gib * GiB) // slot
    row["base=%dGiB" % base_gib] = run(nfr, out[b * per:(b + nfr) * per], 0, 0)
print(json.dumps(row), flush=True)
# fill of the same places, for reference
row = {"torch_fill": True}
for base_gib in (0, 32, 64, 96):
    b = int(base_gib * GiB) // slot
    o = out[b * per:(b + nfr) * per]
    ms = timeit(lambda: o.fill_(1.0), reps=5)
    row["base=%dGiB" % base_gib] = round(o.numel() * 4 / ms / 1e9, 3)
print(json.dumps(row), flush=True)
