"""Round 4: 4-bit contiguous output -- k_decode_flat_lut (product) against k_decode_flat_lds<4>
with direct-to-LDS loads (variant 20), at bench.py's sizes (16 GiB in -> 137 GB out): VDIF
8032-byte frames and GSB 4 MiB blocks; also 1-bit for completeness.  3 rounds interleaved.
    BB_EXPERIMENTS=1 python tools/experiments/exp_glds5.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib
assert _lib.EXPERIMENTS
dev = torch.device('cuda', 0)
kernels.init()
buf = torch.empty((31 << 30) + 4096, dtype=torch.uint8, device=dev)
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


arms = [("lut_product", 5, 0), ("lds_glds_t8", 20, 0), ("lds_glds_t6", 20, 3), ("lds_glds_t4", 20, 2)]
for name, frame, pay, hdr, coder, bps, lim in (("VDIF 4-bit 8032", 8032, 8000, 32, _lib.CODER_VDIF, 4, 16 << 30),
                                                ("GSB 4-bit 4 MiB blocks", 1 << 22, 1 << 22, 0, _lib.CODER_INT, 4, 16 << 30),
                                                ("VDIF 4-bit 8032, 8 GiB in", 8032, 8000, 32, _lib.CODER_VDIF, 4, 8 << 30)):
    nfr = min(lim // frame, out.numel() // (pay * 8 // bps))
    o = out[:nfr * (pay * 8 // bps)]
    res = {a[0]: [] for a in arms}
    for rnd in range(3):
        for label, v, t in arms:
            kernels.tune(_lib.TUNE_FLAT_VARIANT, v); kernels.tune(_lib.TUNE_LUT_TILES, t)
            ms = ms_of(lambda: kernels.decode_frames(buf, nfr, pay, coder, bps, src0=hdr, src_stride=frame, out=o))
            res[label].append(round(nfr * (frame + pay * 8 // bps * 4) / ms / 1e6 / 8000, 4))
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_LUT_TILES, 0)
    print(json.dumps({"case": name, "median": {k: float(np.median(v)) for k, v in res.items()}, "frac_of_8TBps": res}), flush=True)
