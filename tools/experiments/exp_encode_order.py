"""k_encode_flat: the 16 KiB input runs of a launch in input order or dealt over
2^lw stripes (BB_TUNE_ENCODE_STRIPES), 32 and 120 GiB of float32 input; plus a
plain streaming read of the same tensor (torch.sum) for scale.
    python tools/experiments/exp_encode_order.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib, placement          # noqa: E402
from tools.bench_formats import timeit          # noqa: E402

kernels.init()
for gib in (32, 120):
    n = gib * 2 ** 30 // 4
    x = torch.empty(n, dtype=torch.float32, device='cuda')
    for lo in range(0, n, 1 << 28):
        x[lo:lo + (1 << 28)].normal_(0, 2.2)
    ms = timeit(lambda: x.view(-1, 1 << 20).sum(dtype=torch.float32), reps=3)
    print(json.dumps({"case": "torch.sum", "input_GiB": gib, "TBps": round(n * 4 / ms / 1e9, 3)}), flush=True)
    for name, coder, bps in (('vdif', 0, 2), ('vdif', 0, 4), ('vdif', 0, 8), ('vdif', 0, 1)):
        res = {}
        ref = None
        for rnd in range(2):
            for lw in (0, 2, 4, 6):
                kernels.tune(_lib.TUNE_ENCODE_STRIPES, lw)
                out = kernels.encode_flat(x, coder, bps)
                if ref is None:
                    ref = out
                elif rnd == 0:
                    assert torch.equal(out, ref), (name, bps, lw)
                del out
                ms = timeit(lambda: kernels.encode_flat(x, coder, bps), reps=5)
                res.setdefault("lw%d" % lw, []).append(round((n * 4 + n * bps // 8) / ms / 1e9, 3))
        kernels.tune(_lib.TUNE_ENCODE_STRIPES, 0)
        del ref
        print(json.dumps({"case": "%s %d-bit" % (name, bps), "input_GiB": gib, "TBps": res}), flush=True)
    del x
