#!/usr/bin/env python3
"""Experiment: grid size (BB_TUNE_BLOCKS) for the rows / Mark 4 / tiled / flat kernels."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
dev = 'cuda'
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
cases = {}
fn_, pn, nth = 8032, 8000, 8
nsets = nbytes // (fn_ * nth)
perm = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=dev)
pos = torch.arange(nsets, device=dev, dtype=torch.int64)[:, None] * nth + perm[None, :]
src = (pos * fn_ + 32).reshape(-1).contiguous()
out = torch.empty((nbytes // 160000 + 1) * 640000 + nsets * nth * 64, dtype=torch.float32, device=dev)
cases['rows 8x16 complex'] = (lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src,
                                                            complex_data=True, out=out[:nsets * nth * pn * 4]), nsets * nth * fn_ + nsets * nth * pn * 16)
m = BITMAPS[(8, 2, 4)]
nfr = nbytes // 160000
cases['mark4 64 tracks'] = (lambda: kernels.decode_mark4(buf, nfr, 64, 20000, m['sign_bit'], m['mag_bit'], fill_words=160,
                                                         src0=0, src_stride=160000, out=out[:nfr * 20000 * 32]),
                            nfr * 160000 + nfr * 20000 * 32 * 4)
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nb = (nbytes // blk)
g = nb * T * npol * nchan * 2
cases['guppi tiled'] = (lambda: kernels.decode_i8_tiled(buf, nb, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T, src0=0,
                                                        src_stride=blk, out=out[:g]), g * 5)
nf1 = nbytes // 8032
cases['flat cfg1'] = (lambda: kernels.decode_frames(buf, nf1, 8000, 0, 2, src0=32, src_stride=8032, out=out[:nf1 * 32000]),
                      nf1 * 8032 + nf1 * 128000)
for name, (fn, alg) in cases.items():
    for blocks in (0, 49152, 98304, 131072, 163840, 196608):
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        ms = timeit(fn)
        print(json.dumps(dict(case=name, blocks=blocks, ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
