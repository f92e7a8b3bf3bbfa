#!/usr/bin/env python3
"""A/B of the transposing int8 kernel (GUPPI both orders, MKBF heaps): this tree's libbbdecode.so
against a library built from an earlier commit (tools/oldlib/libbbdecode_prev.so), same image, same
output, interleaved launches, HIP events.  usage: python tools/ab_xpose.py [GiB in]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib          # noqa: E402

old = C.CDLL(os.path.join(ROOT, 'tools', 'oldlib', 'libbbdecode_prev.so'))
new = _lib.lib
vp, sz = C.c_void_p, C.c_size_t
old.bb_decode_i8_tiled.restype = C.c_int
old.bb_decode_i8_tiled.argtypes = new.bb_decode_i8_tiled.argtypes
kernels.init()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device('cuda')
npol, nchan = 2, 64
nbytes = int(gib * 2 ** 30)
buf = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
outs = {k: torch.empty(nbytes, dtype=torch.float32, device=dev) for k in ('old', 'new')}
cases = (('GUPPI channels first', _lib.LAYOUT_GUPPI_CF, 128 << 20), ('GUPPI time first', _lib.LAYOUT_GUPPI_TF, 128 << 20),
         ('MKBF heaps', _lib.LAYOUT_MKBF, 256 * 64 * npol * nchan * 2))
for name, layout, blk in cases:
    T = blk // (npol * nchan * 2)
    nfr = nbytes // blk
    p = _lib.TiledParams()
    p.layout, p.npol, p.nchan, p.ntime, p.t_lo, p.t_hi, p.src0, p.src_stride = layout, npol, nchan, T, 0, T, 0, blk
    nelem = nfr * blk
    ts = {'old': [], 'new': []}
    for rep in range(12):
        for k, lib in (('old', old), ('new', new)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.bb_decode_i8_tiled(vp(buf.data_ptr()), buf.numel(), None, nfr, C.byref(p), vp(outs[k].data_ptr()),
                                        nelem, None)
            b.record()
            b.synchronize()
            assert rc == 0, rc
            if rep >= 2:
                ts[k].append(a.elapsed_time(b))
    row = {'case': '%s, %.1f GiB in' % (name, gib), 'identical_output': bool(torch.equal(outs['old'][:nelem], outs['new'][:nelem]))}
    for k in ('old', 'new'):
        ms = float(np.median(ts[k]))
        row[k] = {'ms': round(ms, 4), 'frac_of_8TBps': round(nelem * 5 / ms / 8e9, 4)}
    row['new_over_old'] = round(row['old']['ms'] / row['new']['ms'], 4)
    print(json.dumps(row), flush=True)
