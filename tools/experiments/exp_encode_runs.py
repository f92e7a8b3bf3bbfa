"""k_encode_flat: one or two runs of 256 float4 per wave and step (4 or 8
16-byte loads in flight per lane), 8 and 32 GiB of float32 input.
    python tools/experiments/exp_encode_runs.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402
from tools.bench_formats import timeit          # noqa: E402

kernels.init()
for gib in (8, 32):
    n = gib * 2 ** 30 // 4
    x = torch.empty(n, dtype=torch.float32, device='cuda')
    for lo in range(0, n, 1 << 28):
        x[lo:lo + (1 << 28)].normal_(0, 2.2)
    for name, coder, bps in (('vdif', 0, 1), ('vdif', 0, 2), ('vdif', 0, 4), ('vdif', 0, 8), ('mark5b', 1, 2), ('int', 2, 8)):
        res = {}
        for rnd in range(2):
            for runs in (1, 2):
                kernels.tune(_lib.TUNE_ENCODE_RUNS, runs)
                ms = timeit(lambda: kernels.encode_flat(x, coder, bps), reps=5)
                res.setdefault("runs%d" % runs, []).append(round((n * 4 + n * bps // 8) / ms / 1e9, 3))
        kernels.tune(_lib.TUNE_ENCODE_RUNS, 0)
        print(json.dumps({"case": "%s %d-bit" % (name, bps), "input_GiB": gib, "TBps": res}), flush=True)
    del x
