#!/usr/bin/env python3
"""A/B of the 4-bit encoder: this tree's libbbdecode.so against a library built from an
earlier commit (tools/oldlib/libbbdecode_prev.so: 2-byte stores), same input, same
output buffer, interleaved launches, HIP events.  usage: python tools/ab_encode4.py [GiB]"""
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
for lib in (old,):
    lib.bb_encode_flat.restype = C.c_int
    lib.bb_encode_flat.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
kernels.init()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gib * 2 ** 30) // 4 // 1024 * 1024
x = torch.randn(n, dtype=torch.float32, device='cuda') * 2.2
for coder, name in ((0, 'vdif'), (2, 'int')):
    for bps in (4,):
        outs = {k: torch.zeros(n * bps // 8, dtype=torch.uint8, device='cuda') for k in ('old', 'new')}
        ts = {'old': [], 'new': []}
        for rep in range(12):
            for k, lib in (('old', old), ('new', new)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = lib.bb_encode_flat(C.c_void_p(x.data_ptr()), n, coder, bps, C.c_void_p(outs[k].data_ptr()),
                                        outs[k].numel(), None)
                b.record()
                b.synchronize()
                assert rc == 0, rc
                if rep >= 2:
                    ts[k].append(a.elapsed_time(b))
        same = bool(torch.equal(outs['old'], outs['new']))
        nb = n * 4 + n * bps // 8
        row = {'case': 'encode %s %d-bit, %.1f GiB in' % (name, bps, gib), 'identical_output': same}
        for k in ('old', 'new'):
            ms = float(np.median(ts[k]))
            row[k] = {'ms': round(ms, 4), 'frac_of_8TBps': round(nb / ms / 8e9, 4)}
        row['new_over_old'] = round(row['old']['ms'] / row['new']['ms'], 4)
        print(json.dumps(row), flush=True)
