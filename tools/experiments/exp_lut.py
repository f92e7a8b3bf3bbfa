#!/usr/bin/env python3
"""Byte-table kernel (k_decode_flat_lut) against the register-select kernel
(k_decode_flat_aln), same process, cfg2 layout 2^16 .. 2^20 frames, Mark 5B
and 1-bit layouts, outputs as slices of one big buffer."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nmax = 1 << 20
buf = torch.randint(0, 256, (nmax * 8032 + 8192,), dtype=torch.uint8, device='cuda')
big = torch.empty(nmax * 32000, dtype=torch.float32, device='cuda')
for name, coder, bps, pn, hd in (("cfg2 VDIF 2-bit", _lib.CODER_VDIF, 2, 8000, 32), ("Mark 5B 2-bit", _lib.CODER_MARK5B, 2, 10000, 16),
                                 ("VDIF 1-bit", _lib.CODER_VDIF, 1, 8000, 32), ("VDIF 2-bit 8192-byte payloads", _lib.CODER_VDIF, 2, 8192, 32)):
    for nfr in (1 << 16, 1 << 18, 1 << 19, 1 << 20):
        n = min(nfr, (big.numel() * bps) // (pn * 8), (buf.numel() - 8192) // (pn + hd))
        o = big[:n * pn * 8 // bps]
        fn = lambda: kernels.decode_frames(buf, n, pn, coder, bps, src0=hd, src_stride=pn + hd, out=o)
        row = {"case": name, "frames": n, "out_GB": round(o.numel() * 4 / 1e9, 1)}
        for rep in range(2):
            for lut in (0, 1):
                kernels.tune(_lib.TUNE_BYTE_LUT, lut)
                ms = timeit(fn, reps=5)
                row["%s_%d" % ("lut" if lut else "regs", rep)] = round(n * (pn + hd + pn * 8 // bps * 4) / ms / 1e9, 3)
        row["kernel"] = _lib.last_kernel().split(' grid')[0]
        print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_BYTE_LUT, 1)
