"""Round 4: the LDS gather kernels with direct-to-LDS staging (BB_TUNE_GATHER_GLDS 1,
default) against load + ds_write (0): cfg3 (8 threads x 16 channels complex, 8 GiB),
the sample.vdif layout (8 threads x 1 channel), and a folded subset of 2 of 16 channels.
Interleaved, 3 rounds, digests compared.
    python tools/experiments/exp_gather_glds.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib
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


def case(name, fn, alg, o):
    res = {"glds": [], "regs": []}
    dg = {}
    for rnd in range(3):
        for label, v in (("glds", 1), ("regs", 0)):
            kernels.tune(_lib.TUNE_GATHER_GLDS, v)
            ms = ms_of(fn)
            res[label].append(round(alg / ms / 1e6 / 8000, 4))
            if rnd == 0:
                w = o.view(torch.int32)
                dg[label] = int(w[::1021].to(torch.int64).sum().item())
    kernels.tune(_lib.TUNE_GATHER_GLDS, -1)
    print(json.dumps({"case": name, "kernel": _lib.last_kernel().split(' grid')[0], "frac_of_8TBps": res,
                      "bit_identical": dg["glds"] == dg["regs"],
                      "glds_over_regs": round(float(np.median(res["glds"]) / np.median(res["regs"])), 4)}), flush=True)


nth, fn_, pn = 8, 8032, 8000
nsets = (8 << 30) // (fn_ * nth)
perm = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=dev)
pos = torch.arange(nsets, device=dev, dtype=torch.int64)[:, None] * nth + perm[None, :]
src = (pos * fn_ + 32).reshape(-1).contiguous()
o = out[:nsets * nth * pn * 4]
case("cfg3: 8 threads x 16 channels 2-bit complex", lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=32, nslot=nth, src=src,
     complex_data=True, out=o), nsets * nth * (fn_ + pn * 16), o)
fn2, pn2 = 5032, 5000
nsets2 = min((8 << 30) // (fn2 * nth), out.numel() // (nth * pn2 * 4))
pos2 = torch.arange(nsets2, device=dev, dtype=torch.int64)[:, None] * nth + perm[None, :]
src2 = (pos2 * fn2 + 32).reshape(-1).contiguous()
o2 = out[:nsets2 * nth * pn2 * 4]
case("sample.vdif layout: 8 threads x 1 channel 2-bit real", lambda: kernels.decode_frames(buf, nsets2, pn2, _lib.CODER_VDIF, 2, chunk=1, nslot=nth, src=src2, out=o2),
     nsets2 * nth * (fn2 + pn2 * 16), o2)
within = torch.tensor([6, 7, 24, 25], dtype=torch.int32, device=dev)
o3 = out[:nsets * 1000 * nth * 4]
case("subset 2 of 16 channels folded in", lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=32, nslot=nth, src=src,
     complex_data=True, out=o3, within=within), nsets * nth * fn_ + o3.numel() * 4, o3)
for nth4, ch in ((2, 4), (4, 8), (16, 2)):
    ns = (8 << 30) // (fn_ * nth4)
    s4 = (torch.arange(ns * nth4, device=dev, dtype=torch.int64) * fn_ + 32).contiguous()
    o4 = out[:ns * nth4 * pn * 4]
    case("{} threads x chunk {} 2-bit".format(nth4, ch), lambda: kernels.decode_frames(buf, ns, pn, _lib.CODER_VDIF, 2, chunk=ch, nslot=nth4, src=s4, out=o4),
         ns * nth4 * (fn_ + pn * 16), o4)
