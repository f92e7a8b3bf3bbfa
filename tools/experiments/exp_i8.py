#!/usr/bin/env python3
"""8-bit kernels: old (k_tiled.h) against new (k_xpose.h) forms, with and
without the striped work order, on 4 and 16 GiB inputs (17 / 69 GB of output)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
big = len(sys.argv) > 1 and sys.argv[1] == 'big'
out = torch.empty((17 << 30) if big else (5 << 30), dtype=torch.float32, device=dev)
for gib in ((4, 16) if big else (4,)):
    nbytes = gib << 30
    buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
    npol, nchan, blk = 2, 64, 128 << 20
    T = blk // (npol * nchan * 2)
    nfr = nbytes // blk
    nb = nfr * blk
    o = out[:nb]
    Tm = 256 * 64
    blkm = Tm * npol * nchan * 2
    nfm = nbytes // blkm
    cases = {
        "GUPPI channels-first": lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=o),
        "GUPPI channels-first overlap 512": lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T - 512, src0=0, src_stride=blk, out=o[:nfr * (T - 512) * npol * nchan * 2]),
        "GUPPI time-first": lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=o),
        "MKBF heaps": lambda: kernels.decode_i8_tiled(buf, nfm, _lib.LAYOUT_MKBF, npol, nchan, Tm, 0, Tm, src0=0, src_stride=blkm, out=o[:nfm * blkm]),
        "flat int8 (DADA)": lambda: kernels.decode_frames(buf, 1, nb, _lib.CODER_INT, 8, src0=0, out=o),
        "flat int8 in 8000-byte frames": lambda: kernels.decode_frames(buf, nb // 8032, 8000, _lib.CODER_INT, 8, src0=32, src_stride=8032, out=o[:nb // 8032 * 8000]),
        "VDIF 8-bit in 8000-byte frames": lambda: kernels.decode_frames(buf, nb // 8032, 8000, _lib.CODER_VDIF, 8, src0=32, src_stride=8032, out=o[:nb // 8032 * 8000]),
    }
    for name, fn in cases.items():
        row = {"case": name, "GiB_in": gib}
        for xp in (0, 1):
            for lw in (0, 4):
                kernels.tune(_lib.TUNE_XPOSE, xp)
                kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
                ms = timeit(fn, reps=5)
                moved = nb * 5 if 'overlap' not in name else nfr * (T - 512) * npol * nchan * 2 * 5
                if '8000-byte' in name:
                    moved = nb // 8032 * (8032 + 32000)
                row["xpose%d_stripes%d" % (xp, 1 << lw)] = round(moved / ms / 1e9, 3)
                row["kernel_xpose%d" % xp] = _lib.last_kernel().split(' grid')[0]
        # flat kernels: the other variants too
        if 'flat' in name or 'VDIF' in name:
            for v in (0, 5, 9):
                kernels.tune(_lib.TUNE_FLAT_VARIANT, v)
                kernels.tune(_lib.TUNE_TILES_PER_WAVE_8BIT, 32 if v == 5 else 12)
                for lw in (0, 4):
                    kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
                    ms = timeit(fn, reps=5)
                    row["variant%d_stripes%d" % (v, 1 << lw)] = round(moved / ms / 1e9, 3)
            kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
            kernels.tune(_lib.TUNE_TILES_PER_WAVE_8BIT, 12)
        print(json.dumps(row), flush=True)
    del buf
kernels.tune(_lib.TUNE_XPOSE, 1)
kernels.tune(_lib.TUNE_WORK_STRIPES, 4)
