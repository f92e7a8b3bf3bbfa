"""Round 4: k_copy_frames (DADA NBIT 32 passthrough) -- loads in flight per lane,
non-temporal loads, grid cap, work order; 31 GiB in 128 MiB payloads behind
4096-byte headers, against torch's copy_.  Interleaved, 3 rounds.
    BB_EXPERIMENTS=1 python tools/experiments/exp_copy.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib
assert _lib.EXPERIMENTS
dev = torch.device('cuda', 0)
kernels.init()
blk = 128 << 20
nfr = 247
buf = torch.empty(nfr * (blk + 4096) + 4096, dtype=torch.uint8, device=dev)
buf.view(torch.int32).random_()
out = torch.empty(nfr * blk // 4, dtype=torch.float32, device=dev)


def ms_of(fn, reps=4):
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


arms = [("nl4_nt (product)", 0, 0, -1), ("nl2_nt", 2 | 256, 0, -1), ("nl8_nt", 8 | 256, 0, -1), ("nl16_nt", 16 | 256, 0, -1),
        ("nl4_plainload", 4, 0, -1), ("nl8_plainload", 8, 0, -1),
        ("nl4_nt_grid16k", 0, 16384, -1), ("nl8_nt_grid16k", 8 | 256, 16384, -1), ("nl8_nt_grid128k", 8 | 256, 131072, -1),
        ("nl4_nt_fileorder", 0, 0, 0), ("nl4_nt_4stripes", 0, 0, 2), ("nl4_nt_64stripes", 0, 0, 6)]
res = {a[0]: [] for a in arms}
res["torch_copy"] = []
ref = None
for rnd in range(3):
    for name, cv, blocks, lw in arms:
        kernels.tune(_lib.TUNE_COPY, cv)
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
        ms = ms_of(lambda: kernels.copy_frames(buf, nfr, blk, src0=4096, src_stride=blk + 4096, out=out))
        res[name].append(round(2 * nfr * blk / ms / 1e6 / 8000, 4))
        if rnd == 0:
            d = int(out.view(torch.int32)[::4097].to(torch.int64).sum().item())
            ref = d if ref is None else ref
            assert d == ref, name
    kernels.tune(_lib.TUNE_COPY, 0); kernels.tune(_lib.TUNE_BLOCKS, 0); kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
    src = buf[4096:4096 + (out.numel() * 4)].view(torch.float32)
    ms = ms_of(lambda: out.copy_(src))
    res["torch_copy"].append(round(2 * out.numel() * 4 / ms / 1e6 / 8000, 4))
print(json.dumps({"frac_of_8TBps": res, "median": {k: float(np.median(v)) for k, v in res.items()}}))
