"""Round 4: the 2-bit kernel's defaults (direct-to-LDS loads, 6 tiles per wave) on the
OTHER 2-bit shapes: Mark 5B 10000-byte payloads, VDIF 8192-byte payloads, 5000-byte
payloads, at 8 GiB in (137 GB out), fixed stride.  Arms interleaved, 3 rounds.
    BB_EXPERIMENTS=1 python tools/experiments/exp_glds4.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib
assert _lib.EXPERIMENTS
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


arms = [("glds_t6", 5, 0), ("glds_t4", 5, 4), ("glds_t5", 5, 5), ("regs_t4", 19, 4)]
for name, frame, pay, hdr, coder in (("Mark 5B 10016", 10016, 10000, 16, _lib.CODER_MARK5B), ("VDIF 8224", 8224, 8192, 32, _lib.CODER_VDIF),
                                     ("VDIF 5032", 5032, 5000, 32, _lib.CODER_VDIF), ("VDIF 8032", 8032, 8000, 32, _lib.CODER_VDIF)):
    nfr = min((8 << 30) // frame, out.numel() // (pay * 4))
    o = out[:nfr * pay * 4]
    res = {a[0]: [] for a in arms}
    for rnd in range(3):
        for label, v, t in arms:
            kernels.tune(_lib.TUNE_FLAT_VARIANT, v); kernels.tune(_lib.TUNE_LUT_TILES, t)
            ms = ms_of(lambda: kernels.decode_frames(buf, nfr, pay, coder, 2, src0=hdr, src_stride=frame, out=o))
            res[label].append(round(nfr * (frame + pay * 16) / ms / 1e6 / 8000, 4))
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_LUT_TILES, 0)
    print(json.dumps({"case": name, "median": {k: float(np.median(v)) for k, v in res.items()}, "frac_of_8TBps": res}), flush=True)
