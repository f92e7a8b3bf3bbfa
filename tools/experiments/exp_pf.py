"""Variant 26 of the experiment build: k_decode_flat_lds with READ BURSTS ahead
of the work items (every period a wave reads the input of its next N items into
a scratch corner of LDS; their loads proper should then be served by the
memory-side cache, and HBM sees stores only between two bursts).  Persistent
grids, cfg2 geometry without an index, 2^20 frames -> 125 GiB.
    BB_EXPERIMENTS=1 python tools/experiments/exp_pf.py
Needs tools/experiments/pf_variant26.patch applied (git apply) and both
libraries rebuilt: the variant lost (profiles/r04zp_exp_pf.log) and is not in
the tree.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402

kernels.init()
dev = torch.device('cuda', 0)
FB, PB = 8032, 8000
nfr = 1 << 20
img = torch.empty(nfr * FB + 256, dtype=torch.uint8, device=dev)
for lo in range(0, img.numel(), 1 << 30):
    img[lo:lo + (1 << 30)].random_(0, 256)
out = torch.empty(nfr * PB * 4, dtype=torch.float32, device=dev)
ALG = nfr * PB * 17


def run():
    kernels.decode_frames(img, nfr, PB, _lib.CODER_VDIF, 2, src0=32, src_stride=FB, out=out)


def timed(reps=4):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return min(ts)


def census():
    return int(out[::4097].to(torch.float64).sum().item() * 1000)


kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
kernels.tune(_lib.TUNE_BLOCKS, 0)
base = timed()
ref = census()
print(json.dumps({"case": "product (one item per workgroup)", "ms": round(base, 3), "frac": round(ALG / base / 8e9, 4)}), flush=True)
for blocks in (2048, 4096):
    kernels.tune(_lib.TUNE_BLOCKS, blocks)
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
    ms = timed()
    print(json.dumps({"case": "persistent, no bursts", "blocks": blocks, "ms": round(ms, 3), "frac": round(ALG / ms / 8e9, 4)}), flush=True)
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 26)
    for period_us, items in ((100, 8), (100, 12), (200, 16), (200, 24), (350, 28), (350, 40), (500, 48), (50, 6)):
        it = items if blocks == 2048 else max(2, items // 2)
        kernels.tune(_lib.TUNE_PF_PERIOD, period_us * 100)
        kernels.tune(_lib.TUNE_PF_ITEMS, it)
        out[:1 << 20].zero_()
        ms = timed()
        ok = census() == ref
        print(json.dumps({"case": "bursts", "blocks": blocks, "period_us": period_us, "items": it,
                          "burst_MiB_chipwide": round(blocks * it * 3072 / 2 ** 20, 1),
                          "ms": round(ms, 3), "frac": round(ALG / ms / 8e9, 4), "same_output": ok}), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
kernels.tune(_lib.TUNE_BLOCKS, 0)
