"""the read / write overlap of the headline decode.
Same-process A/B of k_decode_flat_burst (k_burst.h: one loader wave with
direct-to-LDS loads, double-buffered LDS stage of up to 2 x 76 KiB per
workgroup, 3 / 7 / 15 store waves, loader issue optionally clocked to the 100
MHz wall clock) against the product's k_decode_flat_lds on the headline
launch (8 GiB cfg2 image with its index -> 127.5 GiB of output) and at 2^16 /
2^18 frames into arena blocks.  Outputs are compared bit for bit.

    BB_EXPERIMENTS=1 python tools/experiments/exp_burst.py [quick]       (experiment build)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

assert _lib.EXPERIMENTS, "run with BB_EXPERIMENTS=1 (make -C baseband_amd/csrc EXPERIMENTS=1)"
dev = torch.device('cuda', 0)
kernels.init()
quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
FRAME, PAY, HDR = 8032, 8000, 32
nframes = (8 << 30) // FRAME
g = torch.Generator(device=dev)
g.manual_seed(1)
image = torch.empty(nframes * FRAME // 4, dtype=torch.int32, device=dev)
for lo in range(0, image.numel(), 1 << 28):
    hi = min(image.numel(), lo + (1 << 28))
    image[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
image = image.view(torch.uint8)
src = torch.arange(nframes, device=dev, dtype=torch.int64) * FRAME + HDR
none = torch.full_like(src, -1)
out = torch.empty(nframes * PAY * 4, dtype=torch.float32, device=dev)


def setk(on, nbytes=65536, period=0, waves=15, blocks=0):
    kernels.tune(_lib.TUNE_BURST, on)
    kernels.tune(_lib.TUNE_BURST_BYTES, nbytes)
    kernels.tune(_lib.TUNE_BURST_PERIOD, period)
    kernels.tune(_lib.TUNE_BURST_WAVES, waves)
    kernels.tune(_lib.TUNE_BLOCKS, blocks)


def timed(nf, o, s, reps=4):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(image, nf, PAY, _lib.CODER_VDIF, 2, src=s[:nf], out=o)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts)), _lib.last_kernel()


def digest(o, n=1 << 26):
    # a checksum of int32 views of three pieces of the output
    v = o.view(torch.int32)
    m = v.numel()
    return [int(v[k:k + n].to(torch.int64).sum().item()) for k in (0, (m // 2) & ~3, m - n)]


alg = nframes * (FRAME + PAY * 16)
setk(0)
ms0, k0 = timed(nframes, out, src)
d0 = digest(out)
msw, _ = timed(nframes, out, none)
print(json.dumps({"kernel": k0, "ms": round(ms0, 3), "frac": round(alg / ms0 / 1e6 / 8000, 4),
                  "write_only_ms": round(msw, 3), "over_write_only": round(ms0 / msw, 4)}), flush=True)
# (a) alone: k_decode_flat_lds with global_load_lds_dwordx4 (the product since round 4) against
# load + ds_write_b128 (variant 19: round 3's form)
rows = {"glds": [], "regs": []}
for rep in range(4):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
    rows["glds"].append(round(timed(nframes, out, src)[0], 3))
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 19)
    ms18, k18 = timed(nframes, out, src)
    rows["regs"].append(round(ms18, 3))
    same18 = digest(out) == d0
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
print(json.dumps({"variant_19": k18, "interleaved_ms": rows, "bit_identical_digest": same18,
                  "glds_over_regs_speed": round(float(np.median(rows["regs"]) / np.median(rows["glds"])), 4)}), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == 'glds':
    sys.exit(0)
configs = []
for waves in (15, 7):
    for nbytes in ((65536, 32768, 16384, 79360) if not quick else (65536, 32768)):
        for period in ((0, 1000, 2000, 4000) if not quick else (0, 2000)):
            configs.append((waves, nbytes, period, 0))
for blocks in (256, 512, 1024):
    configs.append((7, 16384, 0, blocks))
    configs.append((3, 16384, 0, blocks * 2))
best = None
for waves, nbytes, period, blocks in configs:
    setk(1, nbytes, period, waves, blocks)
    try:
        ms, kn = timed(nframes, out, src)
    except Exception as exc:
        print(json.dumps({"waves": waves, "bytes": nbytes, "period": period, "error": repr(exc)[:200]}), flush=True)
        setk(0)
        continue
    same = digest(out) == d0
    msw1, _ = timed(nframes, out, none, reps=2)
    setk(0)
    row = {"waves": waves, "bytes": nbytes, "period_ticks": period, "blocks": blocks, "ms": round(ms, 3),
           "frac": round(alg / ms / 1e6 / 8000, 4), "vs_product": round(ms0 / ms, 4), "bit_identical_digest": same,
           "write_only_ms": round(msw1, 3), "kernel": kn}
    print(json.dumps(row), flush=True)
    if same and (best is None or ms < best[0]):
        best = (ms, waves, nbytes, period, blocks)
# the product again (drift of the box) and the best configuration, interleaved
if best:
    rows = {"product": [], "burst": []}
    for rep in range(4):
        setk(0)
        rows["product"].append(round(timed(nframes, out, src)[0], 3))
        setk(1, best[2], best[3], best[1], best[4])
        rows["burst"].append(round(timed(nframes, out, src)[0], 3))
    setk(0)
    print(json.dumps({"interleaved_ms": rows, "best": {"waves": best[1], "bytes": best[2], "period": best[3], "blocks": best[4]},
                      "burst_over_product_speed": round(float(np.median(rows["product"]) / np.median(rows["burst"])), 4)}), flush=True)
    # mid sizes into arena blocks
    del out
    torch.cuda.empty_cache()
    ar = arena.Arena(200 << 30)
    for lf in (16, 18):
        nf = 1 << lf
        o = ar.empty(nf * PAY * 4)
        r = {"frames": nf, "product": [], "burst": []}
        for rep in range(3):
            setk(0)
            r["product"].append(round(nf * (FRAME + PAY * 16) / timed(nf, o, src, 6)[0] / 1e6 / 8000, 4))
            setk(1, best[2], best[3], best[1], best[4])
            r["burst"].append(round(nf * (FRAME + PAY * 16) / timed(nf, o, src, 6)[0] / 1e6 / 8000, 4))
        setk(0)
        print(json.dumps(r), flush=True)
        del o
