"""Same-process A/B in the product setting: the cfg2 decode
of 2^15 .. 2^18 frames -- every launch the NEXT window of an 8 GiB image that is
resident in HBM, no input reuse -- into
  A  fresh ``torch.empty`` tensors (one per draw),
  B  tensors from a placement arena (`baseband_amd.arena.Arena`, chunks spread
     over the free HBM and dealt round robin), a fresh block per draw, blocks of
     earlier draws partly held so that draws land at different places,
  C  the same with an arena whose chunks lie wherever the driver put them.
A, B and C take turns.  Output: one JSON line per (size, kind) with the
per-draw rates (TB/s of algorithmic bytes) and min / median / max of the
fraction of 8 TB/s.
    python tools/experiments/exp_arena.py [arena GiB, default 48] [names: comma list of name:0, one arena each]
(rounds r03f / r03g also had arenas whose chunks were steered over the free memory with placeholder
handles, and chunks mapped tooth after tooth instead of dealt: no gain, gone from the library)
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

FRAME, PAYLOAD, HDR = 8032, 8000, 32
IMG_FRAMES = 1 << 20
dev = torch.device('cuda', 0)
kernels.init()
g = torch.Generator(device=dev)
g.manual_seed(1)
image = torch.empty(IMG_FRAMES * FRAME // 4, dtype=torch.int32, device=dev)
for lo in range(0, image.numel(), 1 << 28):
    hi = min(image.numel(), lo + (1 << 28))
    image[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
image = image.view(torch.uint8)
torch.cuda.synchronize()
nxt = [0]


def rate(out, nf, reps=6):
    ts = []
    for r in range(reps + 1):
        if nxt[0] + nf > IMG_FRAMES:
            nxt[0] = 0
        first = nxt[0]
        nxt[0] += nf
        win = image[first * FRAME:(first + nf) * FRAME]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, PAYLOAD, _lib.CODER_VDIF, 2, src0=HDR, src_stride=FRAME, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return nf * (FRAME + PAYLOAD * 16) / float(np.median(ts)) / 1e9


gib = float(sys.argv[1]) if len(sys.argv) > 1 else 48.0
spec = sys.argv[2] if len(sys.argv) > 2 else 'arena:0'
arenas = []
for item in spec.split(','):
    name, mode = item.split(':')
    t0 = time.perf_counter()
    ar = arena.Arena(int(gib * 2 ** 30))
    arenas.append((name, ar))
    print(json.dumps({"arena_GiB": gib, "name": name, "mode": int(mode), "stats": ar.stats(),
                      "wall_s": round(time.perf_counter() - t0, 2)}), flush=True)
DRAWS = 6
for lf in (15, 16, 17, 18):
    nf = 1 << lf
    n = nf * PAYLOAD * 4
    res = {'torch.empty': []}
    res.update({name: [] for name, _ in arenas})
    held = {name: [] for name, _ in arenas}
    for d in range(DRAWS):
        o = torch.empty(n, dtype=torch.float32, device=dev)
        res['torch.empty'].append(round(rate(o, nf), 3))
        del o
        torch.cuda.empty_cache()                     # the next draw is a new allocation
        for name, ar in arenas:
            o = ar.empty(n)
            if o is None:                            # arena full of held blocks: start over
                held[name].clear()
                o = ar.empty(n)
            res[name].append(round(rate(o, nf), 3))
            # keep a small piece so that the next draw starts elsewhere
            held[name].append(ar.empty((64 << 20) // 4))
            del o
    for name, v in res.items():
        f = np.array(v) / 8.0
        print(json.dumps({"log2_frames": lf, "output_GB": round(n * 4 / 1e9, 2), "kind": name, "TBps": v,
                          "frac_min_median_max": [round(float(f.min()), 4), round(float(np.median(f)), 4),
                                                  round(float(f.max()), 4)]}), flush=True)

for name, ar in arenas:
    print(json.dumps({"arena_stats_at_end": name, "stats": ar.stats()}), flush=True)
