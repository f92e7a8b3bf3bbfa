"""Where the time of a mid-size read() goes: fh.read(2^15 frames) on an image
resident in HBM (output allocated by the reader).  Host profile + wall time
against the decode kernel's own duration.
    python tools/prof_mid_read.py
"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from baseband_amd import vdif, kernels, _lib            # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (2 << 30) // bench.FRAME_NBYTES
image, _ = bench.image_buffer(nframes * bench.FRAME_NBYTES, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
SPF = bench.SPF
for nf in (1 << 13, 1 << 15, 1 << 16):
    with vdif.open(image, 'rs', sample_rate=float(SPF * bench.FRAME_RATE)) as fh:
        for k in range(3):
            fh.seek(((k * 3 + 1) * nf % (nframes - nf)) * SPF)
            got = fh.read(nf * SPF)
            del got
        torch.cuda.synchronize()
        ts = []
        pr = cProfile.Profile()
        for k in range(20):
            fh.seek(((k * 7 + 2) * nf % (nframes - nf)) * SPF)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pr.enable()
            got = fh.read(nf * SPF)
            pr.disable()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t0))
            del got
        # back to back, no host sync in between
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            fh.seek(((k * 5 + 3) * nf % (nframes - nf)) * SPF)
            got = fh.read(nf * SPF)
            del got
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("frames %d: 20 reads back to back: host %.3f ms per read, all done %.3f ms per read" %
              (nf, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
        # the decode kernel alone into a fresh arena block of the same size
        out = bench.image_buffer(nf * SPF * 4, dev)[0].view(torch.float32)
        ev = []
        for k in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            kernels.decode_frames(image, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32 + (k * 1000 % (nframes - nf)) * 8032,
                                  src_stride=bench.FRAME_NBYTES, out=out)
            b.record()
            b.synchronize()
            ev.append(a.elapsed_time(b))
        del out
        host = np.median([t[0] for t in ts]) * 1e3
        wall = np.median([t[1] for t in ts]) * 1e3
        print("frames %d: read() returns after %.3f ms, done after %.3f ms; decode kernel alone %.3f ms" %
              (nf, host, wall, float(np.median(ev[1:]))))
        st = pstats.Stats(pr)
        rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:22]
        print("   us/read (own)  us/read (cum)  calls/read  function")
        for (fn, ln, name), (cc, nc, tt, ct, _) in rows:
            print("   %12.1f  %13.1f  %10.1f  %s:%d %s" % (tt / 20 * 1e6, ct / 20 * 1e6, nc / 20, os.path.basename(fn), ln, name))
