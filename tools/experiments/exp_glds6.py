"""Round 4: waves per workgroup of the 2-bit kernel (2 = product; 4: variant 21; 1: variant 22) x tiles per
wave, under bench.py's conditions (image in arena memory, index, 127.5 GiB output).  Interleaved, 3 rounds.
    BB_EXPERIMENTS=1 python tools/experiments/exp_glds6.py"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from baseband_amd import kernels, _lib
assert _lib.EXPERIMENTS
dev = torch.device('cuda', 0)
kernels.init()
FRAME, PAY, HDR = 8032, 8000, 32
nframes = (8 << 30) // FRAME
image, where = bench.image_buffer(nframes * FRAME, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
out = bench.empty_with_patience(nframes * 32000, torch.float32, dev)
src = torch.arange(nframes, device=dev, dtype=torch.int64) * FRAME + HDR


def ms_of(fn, reps=4):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


arms = [("2w_t6", 5, 0), ("4w_t3", 21, 3), ("4w_t4", 21, 4), ("4w_t6", 21, 6), ("1w_t6", 22, 6), ("1w_t8", 22, 8), ("2w_t5", 5, 5), ("2w_t7", 5, 7)]
if len(sys.argv) > 1 and sys.argv[1] == 'aux':
    # cache policy bits of the direct-to-LDS loads: default / sc0 / nt / sc0 nt
    arms = [("default", 5, 0), ("sc0", 23, 0), ("nt", 24, 0), ("sc0_nt", 25, 0)]
res = {a[0]: [] for a in arms}
dg = {}
for rnd in range(3):
    for label, v, t in arms:
        kernels.tune(_lib.TUNE_FLAT_VARIANT, v); kernels.tune(_lib.TUNE_LUT_TILES, t)
        ms = ms_of(lambda: kernels.decode_frames(image, nframes, PAY, _lib.CODER_VDIF, 2, src=src, out=out))
        res[label].append(round(nframes * (FRAME + PAY * 16) / ms / 1e6 / 8000, 4))
        if rnd == 0:
            dg[label] = int(out.view(torch.int32)[::1019].to(torch.int64).sum().item())
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_LUT_TILES, 0)
print(json.dumps({"median": {k: float(np.median(v)) for k, v in res.items()}, "bit_identical": len(set(dg.values())) == 1, "frac_of_8TBps": res}))
