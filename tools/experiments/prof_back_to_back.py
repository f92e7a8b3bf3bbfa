"""Back-to-back reads of 2^15 cfg2 frames from a resident image: host time per
read and time per read, with the scan on the caller's stream (BB_SIDE_SCAN off)
and on a side stream, at the side stream's priorities."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                            # noqa: E402
from baseband_amd import vdif, kernels                  # noqa: E402
from baseband_amd.base import base as bbase             # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (4 << 30) // bench.FRAME_NBYTES
image, _ = bench.image_buffer(nframes * bench.FRAME_NBYTES, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
SPF = bench.SPF
for nf in (1 << 14, 1 << 15, 1 << 16):
    for label, side, prio in (("scan on the caller's stream", False, '-1'), ("side stream, high priority", True, '-1'),
                              ("side stream, normal priority", True, '0')):
        bbase._SIDE_SCAN = side
        os.environ['BB_SIDE_SCAN_PRIORITY'] = prio
        bbase._scan_streams.clear()
        with vdif.open(image, 'rs', sample_rate=float(SPF * bench.FRAME_RATE)) as fh:
            for k in range(4):
                fh.seek(((k * 3 + 1) * nf % (nframes - nf)) * SPF)
                got = fh.read(nf * SPF)
                del got
            torch.cuda.synchronize()
            rows = []
            for rnd in range(3):
                t0 = time.perf_counter()
                for k in range(20):
                    fh.seek(((k * 5 + 3) * nf % (nframes - nf)) * SPF)
                    got = fh.read(nf * SPF)
                    del got
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                rows.append(((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
            # one read alone
            ts = []
            for k in range(8):
                fh.seek(((k * 7 + 2) * nf % (nframes - nf)) * SPF)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                got = fh.read(nf * SPF)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                ts.append((t1 - t0, time.perf_counter() - t0))
                del got
            h, a = min(rows, key=lambda r: r[1])
            alg = nf * (bench.FRAME_NBYTES + SPF * 4)
            print("frames %6d  %-30s back to back: host %.3f ms / read, %.3f ms / read = %.4f of 8 TB/s;  one read: returns after %.3f ms, done after %.3f ms = %.4f"
                  % (nf, label, h, a, alg / a / 1e6 / 8000, np.median([t[0] for t in ts]) * 1e3, np.median([t[1] for t in ts]) * 1e3,
                     alg / (np.median([t[1] for t in ts]) * 1e3) / 1e6 / 8000), flush=True)
