"""Round 4: Mark 4 64-bit words through k_decode_mark4_lds (direct-to-LDS staging; BB_TUNE_M4_LDS 1)
against k_decode_mark4 (register loads + shuffles; 0): 64 tracks fanout 4 at 8 GiB and 2 GiB in,
and the 32- / 16-track modes as super-words; tiles per wave 8 / 4 / 2.  Interleaved, 3 rounds.
    BB_EXPERIMENTS=1 python tools/experiments/exp_m4lds.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
dev = torch.device('cuda', 0)
kernels.init()
buf = torch.empty((8 << 30) + 4096, dtype=torch.uint8, device=dev)
buf.view(torch.int32).random_()
out = torch.empty(34_222_816_000, dtype=torch.float32, device=dev)


def ms_of(fn, reps=4):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


arms = [("lds_t8", 1, 8), ("lds_t4", 1, 4), ("lds_t2", 1, 2), ("shfl_t8", 0, 8)]
for name, key, ntrack, gib in (("64 tracks fanout 4, 8 GiB", (8, 2, 4), 64, 8), ("64 tracks fanout 4, 2 GiB", (8, 2, 4), 64, 2),
                               ):
    m = BITMAPS[key]
    fb = ntrack * 2500
    nfr = min((gib << 30) // fb, out.numel() // (20000 * ntrack // 2))
    o = out[:nfr * 20000 * (ntrack // 2)]
    res = {a[0]: [] for a in arms}
    dg = {}
    for rnd in range(3):
        for label, lds, t in arms:
            kernels.tune(_lib.TUNE_M4_LDS, lds); kernels.tune(_lib.TUNE_M4_TILES, t)
            ms = ms_of(lambda: kernels.decode_mark4(buf, nfr, ntrack, 20000, m['sign_bit'], m['mag_bit'], fill_words=160,
                                                    src0=0, src_stride=fb, out=o))
            res[label].append(round((nfr * fb + o.numel() * 4) / ms / 1e6 / 8000, 4))
            if rnd == 0:
                dg[label] = (int(o.view(torch.int32)[::1013].to(torch.int64).sum().item()), _lib.last_kernel().split(' grid')[0])
    kernels.tune(_lib.TUNE_M4_LDS, 0); kernels.tune(_lib.TUNE_M4_TILES, 8)
    print(json.dumps({"case": name, "median": {k: float(np.median(v)) for k, v in res.items()},
                      "bit_identical": len({d[0] for d in dg.values()}) == 1, "kernels": {k: d[1] for k, d in dg.items()}}), flush=True)
