#!/usr/bin/env python3
"""Striped work order (bb_perm_t, BB_TUNE_WORK_STRIPES) against file order for
every decode kernel family at 2^16 .. 2^20 frames; outputs are slices of one
big buffer and fresh exactly-sized allocations."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
payload, header = 8000, 32
stride = payload + header
nmax = 1 << 20
buf = torch.randint(0, 256, (nmax * stride + 8192,), dtype=torch.uint8, device=dev)
out = torch.empty(nmax * payload * 4, dtype=torch.float32, device=dev)
per = payload * 4
m = BITMAPS[(8, 2, 4)]
perm8 = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=dev)


def cases(nfr, o):
    nsets = nfr // 8
    pos = torch.arange(nsets, device=dev, dtype=torch.int64)[:, None] * 8 + perm8[None, :]
    src8 = (pos * stride + header).reshape(-1).contiguous()
    nf4 = nfr * per // 640000
    return {
        "flat (cfg2)": (lambda: kernels.decode_frames(buf, nfr, payload, 0, 2, src0=header, src_stride=stride, out=o), nfr * (stride + payload * 16)),
        "rows 8 thr x 16 ch complex (cfg3)": (lambda: kernels.decode_frames(buf, nsets, payload, 0, 2, chunk=32, nslot=8, src=src8, complex_data=True, out=o), nsets * 8 * (stride + payload * 16)),
        "gather 8 thr x 1 ch": (lambda: kernels.decode_frames(buf, nsets, payload, 0, 2, chunk=1, nslot=8, src=src8, out=o), nsets * 8 * (stride + payload * 16)),
        "mark4 64 tracks": (lambda: kernels.decode_mark4(buf, nf4, 64, 20000, m['sign_bit'], m['mag_bit'], fill_words=160, src0=0, src_stride=160000, out=o[:nf4 * 640000]), nf4 * (160000 + 640000 * 4)),
    }


for nfr in (1 << 16, 1 << 18, 1 << 19, 1 << 20):
    for where in ("slice of the 134 GB buffer", "fresh exactly-sized"):
        if where.startswith("fresh"):
            if nfr == nmax:
                continue
            o = torch.empty(nfr * per, dtype=torch.float32, device=dev)
        else:
            o = out[:nfr * per]
        for name, (fn, moved) in cases(nfr, o).items():
            row = {"frames": nfr, "where": where, "case": name}
            for lw in (0, 2, 4, 6):
                kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
                ms = timeit(fn, reps=5)
                row["stripes%d" % (1 << lw)] = round(moved / ms / 1e9, 3)
            row["kernel"] = _lib.last_kernel().split(' grid')[0]
            if name.startswith("flat"):
                for v in (0,):
                    kernels.tune(_lib.TUNE_FLAT_VARIANT, v)
                    for lw in (0, 4):
                        kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
                        ms = timeit(fn, reps=5)
                        row["plain_kernel_stripes%d" % (1 << lw)] = round(moved / ms / 1e9, 3)
                kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
            print(json.dumps(row), flush=True)
        del o
kernels.tune(_lib.TUNE_WORK_STRIPES, 4)
