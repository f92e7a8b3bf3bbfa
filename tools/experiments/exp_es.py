#!/usr/bin/env python3
"""One-pass striped decode (k_decode_flat_es, variants 10-13 = 2/4/8/16 tiles
per wave) against the persistent (5) and plain (0) kernels, cfg2 layout,
2^14 .. 2^20 frames, outputs in fresh exactly-sized allocations and as slices
of one 134 GB buffer; Mark 5B and 8-bit layouts at a few sizes."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
payload, header = 8000, 32
stride = payload + header
nmax = 1 << 20
buf = torch.randint(0, 256, (nmax * stride + 8192,), dtype=torch.uint8, device='cuda')
per = payload * 4


def run(fn, moved, v):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, v)
    ms = timeit(fn, reps=5)
    return round(moved / ms / 1e9, 3)


for where in ("fresh", "slice"):
    big = torch.empty(nmax * per, dtype=torch.float32, device='cuda') if where == "slice" else None
    for nfr in (1 << 14, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20):
        if where == "fresh" and nfr == nmax:
            continue
        o = big[:nfr * per] if big is not None else torch.empty(nfr * per, dtype=torch.float32, device='cuda')
        row = {"frames": nfr, "where": where, "out_GB": round(nfr * per * 4 / 1e9, 1)}
        fn = lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header, src_stride=stride, out=o)
        for v in (5, 0, 10, 11, 14):
            row["v%d" % v] = run(fn, nfr * (stride + payload * 16), v)
        src = torch.arange(nfr, device='cuda', dtype=torch.int64) * stride + header
        fn2 = lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src=src, out=o)
        for v in (5, 14):
            row["indexed_v%d" % v] = run(fn2, nfr * (stride + payload * 16), v)
        print(json.dumps(row), flush=True)
        del o
    if big is not None:
        # other layouts on slices of the big buffer
        for name, coder, bps, pn, hd in (("Mark 5B 10000-byte payloads", _lib.CODER_MARK5B, 2, 10000, 16),
                                         ("int8 8000-byte payloads", _lib.CODER_INT, 8, 8000, 32),
                                         ("VDIF 4-bit", _lib.CODER_VDIF, 4, 8000, 32), ("VDIF 1-bit", _lib.CODER_VDIF, 1, 8000, 32)):
            for nfr in (1 << 16, 1 << 19):
                n = min(nfr, (big.numel() * bps) // (pn * 8))
                o = big[:n * pn * 8 // bps]
                row = {"case": name, "frames": n, "out_GB": round(o.numel() * 4 / 1e9, 1)}
                fn = lambda: kernels.decode_frames(buf, n, pn, coder, bps, src0=hd, src_stride=pn + hd, out=o)
                for v in (5, 0, 11, 14):
                    row["v%d" % v] = run(fn, n * (pn + hd + pn * 8 // bps * 4), v)
                print(json.dumps(row), flush=True)
    del big
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
