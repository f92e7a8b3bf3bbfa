"""What stands in front of the decode kernel in a mid-size read(): HIP-event times of
the scan, the index + verification and the decode launches of one 2^15-frame cfg2
window (the three launches of bb_vdif_read_window), the decode with and without an
index, and the same window through fh.read().
    python tools/prof_read_breakdown.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from baseband_amd import vdif, kernels, _lib            # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (2 << 30) // bench.FRAME_NBYTES
image, _ = bench.image_buffer(nframes * bench.FRAME_NBYTES, dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
pattern, mask = h0.invariant_pattern()
SPF, FN, PN = bench.SPF, bench.FRAME_NBYTES, bench.PAYLOAD_NBYTES


def ev():
    return torch.cuda.Event(enable_timing=True)


for nf in (1 << 13, 1 << 15, 1 << 16):
    out = bench.image_buffer(nf * SPF * 4, dev)[0].view(torch.float32)
    rows = []
    for k in range(8):
        first = (k * 4099 + 7) % (nframes - nf)
        win = image[first * FN:(first + nf) * FN]
        e = [ev() for _ in range(5)]
        e[0].record()
        recs = kernels.vdif_scan(win, nf, FN, 32, pattern, mask, h0['seconds'], h0['frame_nr'] + first, bench.FRAME_RATE)
        e[1].record()
        src = kernels.build_index(recs, nf, 1, None)
        e[2].record()
        kernels.decode_frames(win, nf, PN, _lib.CODER_VDIF, 2, src=src, out=out)
        e[3].record()
        kernels.decode_frames(win, nf, PN, _lib.CODER_VDIF, 2, src0=32, src_stride=FN, out=out)
        e[4].record()
        e[4].synchronize()
        rows.append([e[i].elapsed_time(e[i + 1]) * 1e3 for i in range(4)])
    med = np.median(np.array(rows[2:]), axis=0)
    print("frames %6d: scan %.1f us, index %.1f us, decode with index %.1f us, decode fixed stride %.1f us "
          "(event to event on one stream: launch gaps included)" % ((nf,) + tuple(med)))
    del out
    with vdif.open(image, 'rs', sample_rate=float(SPF * bench.FRAME_RATE)) as fh:
        ts = []
        for k in range(12):
            fh.seek(((k * 7 + 2) * nf % (nframes - nf)) * SPF)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = fh.read(nf * SPF)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t0))
            del got
        ts = np.array(ts[3:]) * 1e6
        print("frames %6d: fh.read(): returns after %.1f us, done after %.1f us (medians); kernel name %s" %
              (nf, np.median(ts[:, 0]), np.median(ts[:, 1]), _lib.last_kernel()))
