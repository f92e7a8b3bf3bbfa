"""k_decode_i8_xpose: channels per tile for blocks of 64 channels.  A narrower
tile is longer along the input's contiguous axis (the tile keeps its 16 KiB of
input): TC = 64 reads 256-byte runs of a channels-first block (128 bytes of an
MKBF heap row with two pols), TC = 32 twice that, TC = 16 four times -- and
writes output rows in 512 / 256 / 128-byte pieces.  8 GiB and 31 GiB of input,
outputs in the arena (8 GiB) or a plain tensor (31 GiB: 133 GB).
    python tools/experiments/exp_xpose_tc.py
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
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)


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
    nfr = nbytes // blk
    out = ar.empty(nbytes) if where == 'arena' else torch.empty(nbytes, dtype=torch.float32, device=dev)
    for layout, name in ((_lib.LAYOUT_GUPPI_CF, "GUPPI channels first"), (_lib.LAYOUT_GUPPI_TF, "GUPPI time first"),
                         (_lib.LAYOUT_MKBF, "MKBF heaps")):
        res = {}
        for rnd in range(2):
            for tc in (64, 32, 16, 8):
                for rows in (128, 64):
                    kernels.tune(_lib.TUNE_XPOSE_TC, tc)
                    kernels.tune(_lib.TUNE_XPOSE_ROWS, rows)
                    r = rate(lambda: kernels.decode_i8_tiled(buf, nfr, layout, npol, nchan, T, 0, T, src0=0,
                                                             src_stride=blk, out=out), nbytes)
                    res.setdefault("tc{}_rows{}".format(tc, rows), []).append(r)
        kernels.tune(_lib.TUNE_XPOSE_TC, 0)
        kernels.tune(_lib.TUNE_XPOSE_ROWS, 0)
        print(json.dumps({"case": name, "input_GiB": nbytes >> 30, "output": where, "TBps": res}), flush=True)
    del out
