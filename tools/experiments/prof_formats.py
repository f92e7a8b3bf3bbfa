#!/usr/bin/env python3
"""Driver for per-kernel profiles (run under rocprofv3, program directly after
``--``): launches every decode kernel family `reps` times on `gib` GiB of
random input resident in HBM.  usage: prof_formats.py [gib] [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from baseband_amd.mark4._bitmaps import BITMAPS
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kernels.init()
dev = torch.device('cuda')
nbytes = int(gib * 2 ** 30)
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device=dev)
out = torch.empty(nbytes * 17 // 4 + (1 << 20), dtype=torch.float32, device=dev)
stride, payload, header = 8032, 8000, 32
nfr = nbytes // stride
nsets = nfr // 8
perm8 = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=dev)
pos = torch.arange(nsets, device=dev, dtype=torch.int64)[:, None] * 8 + perm8[None, :]
src8 = (pos * stride + header).reshape(-1).contiguous()
m = BITMAPS[(8, 2, 4)]
nf4 = nbytes // 160000
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nfg = max(1, nbytes // blk)
Tm = 256 * 64
blkm = Tm * npol * nchan * 2
nfm = nbytes // blkm
nb = nbytes // 4 * 4
sel8 = torch.tensor([6, 7, 8, 9, 10, 11, 12, 13], dtype=torch.int32, device=dev)
sel5 = torch.arange(8, dtype=torch.int32, device=dev)
src5 = torch.arange(nbytes // 10016, device=dev, dtype=torch.int64) * 10016 + 16
cm32 = torch.arange(0, 64, 2, dtype=torch.int32, device=dev)
m4s = kernels.mark4_select_maps(m['sign_bit'], m['mag_bit'], 8, [0, 5, 7])
runs = [
    lambda: kernels.decode_frames(buf, nfr, payload, 0, 2, src0=header, src_stride=stride, out=out),
    lambda: kernels.decode_frames(buf, nsets, payload, 0, 2, chunk=32, nslot=8, src=src8, complex_data=True, out=out),
    lambda: kernels.decode_frames(buf, nsets, payload, 0, 2, chunk=1, nslot=8, src=src8, out=out),
    lambda: kernels.decode_frames(buf, nbytes // 10016, 10000, _lib.CODER_MARK5B, 2, chunk=16, src0=16, src_stride=10016, out=out),
    lambda: kernels.decode_mark4(buf, nf4, 64, 20000, m['sign_bit'], m['mag_bit'], fill_words=160, src0=0, src_stride=160000, out=out),
    lambda: kernels.decode_i8_tiled(buf, nfg, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=out),
    lambda: kernels.decode_i8_tiled(buf, nfg, _lib.LAYOUT_GUPPI_TF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=out),
    lambda: kernels.decode_i8_tiled(buf, nfm, _lib.LAYOUT_MKBF, npol, nchan, Tm, 0, Tm, src0=0, src_stride=blkm, out=out),
    lambda: kernels.decode_frames(buf, 1, nb, _lib.CODER_INT, 8, src0=0, out=out),
    # channel selections folded into the decode: 4 of 16 complex channels of 8
    # threads, 8 of 16 Mark 5B channels, 3 of 8 Mark 4 channels
    lambda: kernels.decode_frames(buf, nsets, payload, 0, 2, chunk=32, nslot=8, src=src8, complex_data=True,
                                  out=out, within=sel8),
    lambda: kernels.decode_frames(buf, nbytes // 10016, 10000, _lib.CODER_MARK5B, 2, chunk=16, src=src5,
                                  out=out, within=sel5),
    lambda: kernels.decode_mark4(buf, nf4, 64, 20000, m4s[0], m4s[1], fill_words=160, src0=0, src_stride=160000,
                                 out=out, select=True),
    # round 3: 1- and 4-bit flat (k_decode_flat_lut; `gib` / 2 of input for 1-bit so that the output fits),
    # channel list / single polarisation folded into the int8 transposes
    lambda: kernels.decode_frames(buf, nfr // 2, payload, 0, 1, src0=header, src_stride=stride, out=out),
    lambda: kernels.decode_frames(buf, nfr, payload, 0, 4, src0=header, src_stride=stride, out=out),
    lambda: kernels.decode_i8_tiled(buf, nfg, _lib.LAYOUT_GUPPI_CF, npol, 32, T, 0, T, src0=0, src_stride=blk, out=out,
                                    nchan_stored=nchan, npol_stored=npol, chan_map=cm32),
    lambda: kernels.decode_i8_tiled(buf, nfg, _lib.LAYOUT_GUPPI_TF, 1, nchan, T, 0, T, src0=0, src_stride=blk, out=out,
                                    nchan_stored=nchan, npol_stored=npol, pol_first=1),
    lambda: kernels.decode_i8_tiled(buf, nfm, _lib.LAYOUT_MKBF, 1, 32, Tm, 0, Tm, src0=0, src_stride=blkm, out=out,
                                    nchan_stored=nchan, npol_stored=npol, pol_first=0, chan_map=cm32),
]
if os.environ.get('BB_PROF_OLD_I8'):
    kernels.tune(_lib.TUNE_XPOSE, 0)
for fn in runs:
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(_lib.last_kernel(), flush=True)
