#!/usr/bin/env python3
"""k_decode_i8_xpose: grid size (persistent, pipelined <-> one tile per
workgroup) on 16 GiB of GUPPI channels-first / time-first / MKBF input."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
gib = 16
nbytes = gib << 30
out = torch.empty(17 << 30, dtype=torch.float32, device=dev)
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
    "GUPPI time-first": lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=o),
    "MKBF heaps": lambda: kernels.decode_i8_tiled(buf, nfm, _lib.LAYOUT_MKBF, npol, nchan, Tm, 0, Tm, src0=0, src_stride=blkm, out=o),
    "flat int8": lambda: kernels.decode_frames(buf, 1, nb, _lib.CODER_INT, 8, src0=0, out=o),
}
for name, fn in cases.items():
    row = {"case": name}
    for rows in (128, 64):
        kernels.tune(_lib.TUNE_XPOSE_ROWS, rows)
        for blocks in (0, 1 << 17, 1 << 19, 1 << 20):
            kernels.tune(_lib.TUNE_BLOCKS, blocks)
            for lw in (0, 4):
                kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
                ms = timeit(fn, reps=5)
                row["rows%d_blocks%d_stripes%d" % (rows, blocks, 1 << lw)] = round(nb * 5 / ms / 1e9, 3)
    kernels.tune(_lib.TUNE_BLOCKS, 0)
    kernels.tune(_lib.TUNE_XPOSE_ROWS, 128)
    kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
    row["kernel"] = _lib.last_kernel()
    print(json.dumps(row), flush=True)
t = timeit(lambda: o.fill_(1.0), reps=5)
print(json.dumps({"torch_fill_TBps": round(nb * 4 / t / 1e9, 3)}))
i8 = buf[:nb].view(torch.int8)
t = timeit(lambda: torch.Tensor.copy_(o, i8), reps=5)
print(json.dumps({"torch_int8_to_float_copy_TBps": round(nb * 5 / t / 1e9, 3)}))
