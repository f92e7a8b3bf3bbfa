#!/usr/bin/env python3
"""A/B of the byte-granular header search: this tree's libbbdecode.so against a library built from
an earlier commit (tools/oldlib/libbbdecode_prev.so), same image, interleaved launches, HIP events;
the offsets found must be the same set.  usage: python tools/ab_locate.py [GiB]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402
from baseband_amd import kernels, _lib          # noqa: E402

old = C.CDLL(os.path.join(ROOT, 'tools', 'oldlib', 'libbbdecode_prev.so'))
new = _lib.lib
vp, sz = C.c_void_p, C.c_size_t
old.bb_vdif_locate.restype = C.c_int
old.bb_vdif_locate.argtypes = [vp, sz, vp, vp, sz, vp, vp]
old.bb_mark5b_locate.restype = C.c_int
old.bb_mark5b_locate.argtypes = [vp, sz, vp, sz, vp, vp]
kernels.init()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device('cuda')
nframes = int(gib * 2 ** 30) // bench.FRAME_NBYTES
image, h0 = bench.make_file_image_on_device(nframes, 31, 0, dev)
nbytes = image.numel() * image.element_size()
pattern, mask = h0.invariant_pattern()
params = kernels._vdif_params(bench.FRAME_NBYTES, 32, pattern, mask, 0, 0, 0)
cap = nframes + 16


def run(lib, which):
    offs = torch.empty(cap, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if which == 'vdif':
        rc = lib.bb_vdif_locate(vp(image.data_ptr()), nbytes, C.byref(params), vp(offs.data_ptr()), cap,
                                vp(count.data_ptr()), None)
    else:
        rc = lib.bb_mark5b_locate(vp(image.data_ptr()), nbytes, vp(offs.data_ptr()), cap, vp(count.data_ptr()), None)
    b.record()
    b.synchronize()
    assert rc == 0, rc
    n = int(count.item())
    return a.elapsed_time(b), torch.sort(offs[:min(n, cap)]).values


for which in ('vdif', 'mark5b'):
    ts, found = {'old': [], 'new': []}, {}
    for rep in range(12):
        for k, lib in (('old', old), ('new', new)):
            ms, offs = run(lib, which)
            found[k] = offs
            if rep >= 2:
                ts[k].append(ms)
    row = {'case': '%s locate, %.1f GiB' % (which, gib), 'same_offsets': bool(torch.equal(found['old'], found['new'])),
           'found': int(found['new'].numel())}
    for k in ('old', 'new'):
        ms = float(np.median(ts[k]))
        row[k] = {'ms': round(ms, 4), 'frac_of_8TBps': round(nbytes / ms / 8e9, 4)}
    row['new_over_old'] = round(row['old']['ms'] / row['new']['ms'], 4)
    print(json.dumps(row), flush=True)

# the new library alone, over grids (BB_TUNE_BLOCKS: 0 = the default cap of 65536 workgroups)
if os.environ.get('BB_AB_LOCATE_GRIDS'):
    for blocks in (0, 16384, 32768, 131072, 262144, 524288, 1048576):
        new.bb_tune(_lib.TUNE_BLOCKS, blocks)
        for which in ('vdif', 'mark5b'):
            t = []
            for rep in range(10):
                ms, _ = run(new, which)
                if rep >= 2:
                    t.append(ms)
            ms = float(np.median(t))
            print(json.dumps({'case': '%s locate, %.1f GiB, workgroups at most %d' % (which, gib, blocks), 'ms': round(ms, 4),
                              'frac_of_8TBps': round(nbytes / ms / 8e9, 4)}), flush=True)
    new.bb_tune(_lib.TUNE_BLOCKS, 0)
