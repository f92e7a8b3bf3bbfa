"""Contiguous 8-bit output: k_decode_flat_lds<8> (16-byte loads staged in LDS)
against k_decode_flat<8> (a dword per lane), same process, same buffers: DADA
int8 and VDIF 8-bit, 8 GiB in (arena output) and 31 GiB in (plain output).
    python tools/experiments/exp_flat8.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, arena          # noqa: E402

dev = torch.device('cuda')
kernels.init()
big = 31 << 30
buf = torch.empty(big + 4096, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev)
g.manual_seed(3)
v = buf[:big].view(torch.int32)
for lo in range(0, v.numel(), 1 << 28):
    hi = min(v.numel(), lo + (1 << 28))
    v[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
ar = arena.Arena(200 << 30)


def rate(fn, nbytes, reps=5):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return round(5 * nbytes / float(np.median(ts)) / 1e9, 3)


for nbytes, where in ((8 << 30, 'arena'), (big, 'plain')):
    out = ar.empty(nbytes) if where == 'arena' else torch.empty(nbytes, dtype=torch.float32, device=dev)
    for name, coder, pn, hdr in (("DADA int8, 128 MiB frames", _lib.CODER_INT, (128 << 20) - 4096, 4096),
                                 ("VDIF 8-bit, 8032-byte frames", _lib.CODER_VDIF, 8000, 32),
                                 ("VDIF 8-bit, 8224-byte frames", _lib.CODER_VDIF, 8192, 32)):
        nfr = nbytes // (pn + hdr)
        res = {}
        for rnd in range(3):
            for knob, tiles, label in ((1, 4, "staged_16"), (1, 3, "staged_12"), (1, 2, "staged_8"), (1, 1, "staged_4"), (0, 4, "plain")):
                kernels.tune(_lib.TUNE_FLAT8_LDS, knob)
                kernels.tune(_lib.TUNE_LUT_TILES, tiles)
                r = rate(lambda: kernels.decode_frames(buf, nfr, pn, coder, 8, src0=hdr, src_stride=pn + hdr, out=out),
                         nfr * (pn + hdr) * 1.0 / 5 + nfr * pn * 4 / 5)
                res.setdefault(label, []).append(r)
                res[label + "_kernel"] = _lib.last_kernel().split(' grid')[0]
        kernels.tune(_lib.TUNE_FLAT8_LDS, 1)
        kernels.tune(_lib.TUNE_LUT_TILES, 4)
        print(json.dumps({"case": name, "input_GiB": nbytes >> 30, "output": where, "TBps": res}), flush=True)
    del out
