#!/usr/bin/env python3
"""GUPPI / MKBF tiled int8 decode: tile size and grid sweep (kernel only)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nbytes, dtype=torch.float32, device='cuda')
for layout, name, npol, nchan in ((_lib.LAYOUT_MKBF, 'mkbf', 2, 64), (_lib.LAYOUT_MKBF, 'mkbf_1024ch', 2, 1024)):
    blocsize = 128 << 20
    ntime = blocsize // (npol * nchan * 2)
    if layout == _lib.LAYOUT_MKBF:
        ntime = ntime // 256 * 256
    stride = blocsize + 6400
    nfr = nbytes // stride
    alg = nfr * (blocsize + blocsize * 4)
    for te, stage, mtc in ((8192, 0, 32), (4096, 1, 16), (4096, 1, 32), (8192, 1, 16), (8192, 1, 32), (8192, 1, 64), (16384, 1, 32), (16384, 1, 64)):
        kernels.tune(_lib.TUNE_MKBF_CHANNELS, mtc)
        kernels.tune(_lib.TUNE_TILE_ELEMS, te)
        kernels.tune(_lib.TUNE_TILED_STAGE, stage)
        row = dict(layout=name, tile_elems=te, stage=stage, mkbf_tc=mtc)
        for blocks in (0, 131072):
            kernels.tune(_lib.TUNE_BLOCKS, blocks)
            try:
                ms = timeit(lambda: kernels.decode_i8_tiled(buf, nfr, layout, npol, nchan, ntime, 0, ntime,
                                                            src0=6400, src_stride=stride,
                                                            out=out[:nfr * ntime * npol * nchan * 2]), reps=5)
                row['b%d' % blocks] = round(alg / ms / 1e9, 2)
            except Exception as exc:
                row['b%d' % blocks] = type(exc).__name__
        print(json.dumps(row), flush=True)
kernels.tune(_lib.TUNE_TILE_ELEMS, 8192); kernels.tune(_lib.TUNE_BLOCKS, 0); kernels.tune(_lib.TUNE_TILED_STAGE, 1)
