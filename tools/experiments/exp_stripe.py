#!/usr/bin/env python3
"""Is the launch-size cliff a matter of how far over the physical address space
a launch's output is spread?  The same 2^16 .. 2^19-frame launches (cfg2
layout, default kernel) with their output frames dealt over W regions that
lie S frame-slots apart inside one big (2^20-frame, 134 GB) output buffer:
S = 2^20 / W spreads every launch over the whole buffer (what the 2^20-frame
launch does by itself), S = nframes / W only permutes inside the launch's own
range (many windows, no extra spread).  A/B inside one process."""
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


def run(nfr, o, variant, W=0, S=0):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    kernels.tune(_lib.TUNE_OUT_STRIPE_W, W)
    kernels.tune(_lib.TUNE_OUT_STRIPE_S, S)
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                              src_stride=stride, out=o), reps=5)
    kernels.tune(_lib.TUNE_OUT_STRIPE_W, 0)
    return round(nfr * (stride + payload * 16) / ms / 1e9, 3)


# correctness of the striped placement: frame fs lands in slot (fs % W) * S + fs // W
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
ref = kernels.decode_frames(buf, 4096, payload, _lib.CODER_VDIF, 2, src0=header, src_stride=stride).view(4096, per)
kernels.tune(_lib.TUNE_OUT_STRIPE_W, 16)
kernels.tune(_lib.TUNE_OUT_STRIPE_S, 1000)
got = kernels.decode_frames(buf, 4096, payload, _lib.CODER_VDIF, 2, src0=header, src_stride=stride,
                            out=out[:16 * 1000 * per]).view(16000, per)
fs = torch.arange(4096, device='cuda')
assert torch.equal(got[(fs % 16) * 1000 + fs // 16].view(torch.int32), ref.view(torch.int32))
kernels.tune(_lib.TUNE_OUT_STRIPE_W, 0)
print(json.dumps({"striped placement correct": True}), flush=True)

for nfr in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20):
    row = {"frames": nfr}
    for v, name in ((5, "persistent"), (0, "plain")):
        row[name + "_contiguous_at_0"] = run(nfr, out[:nfr * per], v)
        if nfr < nmax:
            row[name + "_contiguous_at_end"] = run(nfr, out[(nmax - nfr) * per:], v)
        for W in (4, 16, 64, 256):
            if nfr < nmax:
                row["%s_W%d_spread_over_134GB" % (name, W)] = run(nfr, out, v, W, nmax // W)
            row["%s_W%d_inside_own_range" % (name, W)] = run(nfr, out[:nfr * per], v, W, nfr // W)
    print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
