"""How much faster do the int8 kernels (reads are a fifth of their traffic) run when their input comes
out of the memory-side cache?  GUPPI channels-first / time-first and flat DADA int8, 1 / 2 blocks of
128 MiB: the same window decoded again and again against another window every time; and a plain
read-through (bb_touch) followed by the decode of another window every time, on one stream."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib, arena          # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nbytes = 4 << 30
buf = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
ar = arena.enable()


def timed(fn, reps=9):
    ts = []
    for r in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(r)
        b.record()
        b.synchronize()
        if r >= 2:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for nfr in (1, 2):
    n_in = nfr * blk
    out = ar.empty((n_in,))
    nwin = nbytes // n_in - 1
    for name, layout in (('GUPPI channels first', _lib.LAYOUT_GUPPI_CF), ('GUPPI time first', _lib.LAYOUT_GUPPI_TF), ('DADA int8 flat', None)):
        def dec(win):
            if layout is None:
                kernels.decode_frames(win, 1, n_in, _lib.CODER_INT, 8, src0=0, out=out)
            else:
                kernels.decode_i8_tiled(win, nfr, layout, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=out)
        same = timed(lambda r: dec(buf[:n_in]))
        other = timed(lambda r: dec(buf[((r * 3 + 1) % nwin) * n_in:][:n_in]))

        def touched(r):
            win = buf[((r * 3 + 2) % nwin) * n_in:][:n_in]
            _lib.lib.bb_touch(C.c_void_p(win.data_ptr()), n_in, None)
            dec(win)
        both = timed(touched)
        print("%-22s %d MiB in: same window %.1f us, another window %.1f us (x%.3f), read through + another window %.1f us (x%.3f)"
              % (name, n_in >> 20, same * 1e3, other * 1e3, other / same, both * 1e3, other / both), flush=True)
