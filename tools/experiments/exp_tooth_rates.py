"""Does the decode rate depend on WHICH gigabyte of physical memory the output lies in?
An arena that grows in steps of 1 GiB (every step one hipMemCreate batch: one "tooth" of
the product's 48 GiB steps), one 1 GiB block per step, a cfg2 decode of 8000 frames
(1.02 GB of output) into each, median of 5 launches; then the same frames into blocks DEALT
over all teeth the way the product maps a step (a 48 GiB step, blocks of 1 GiB and 4 GiB).
    python tools/experiments/exp_tooth_rates.py [GiB, default 64]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GIB = 1 << 30
n_gib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
os.environ['BB_ARENA_STEP_GIB'] = '1'
os.environ['BB_ARENA_RETRY_BELOW_GBPS'] = '0'
import bench                                            # noqa: E402
from baseband_amd import arena, kernels, _lib           # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (2 << 30) // bench.FRAME_NBYTES
image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)


def rate(out, nf, first):
    ts = []
    for r in range(6):
        win = image[((first + r * nf) % (nframes - nf)) * bench.FRAME_NBYTES:][:nf * bench.FRAME_NBYTES]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        kernels.decode_frames(win, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32, src_stride=bench.FRAME_NBYTES, out=out)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / float(np.median(ts)) / 1e6


ar = arena.Arena((n_gib + 2) * GIB)
blocks, rates = [], []
nf = 8000
for k in range(n_gib):
    t = ar.empty(GIB // 4)
    blocks.append(t)
    rates.append(rate(t[:nf * bench.SPF], nf, 17 * k))
r = np.array(rates)
print("teeth of 1 GiB, %d of them: GB/s min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (
    len(r), r.min(), np.percentile(r, 10), np.median(r), np.percentile(r, 90), r.max()))
print("per tooth:", " ".join("%.0f" % x for x in r))
# again, in another order: is a tooth's rate its own?
again = np.array([rate(blocks[k][:nf * bench.SPF], nf, 5 * k + 3) for k in range(n_gib)])
print("again    :", " ".join("%.0f" % x for x in again))
print("correlation of the two passes: %.3f" % np.corrcoef(r, again)[0, 1])
del blocks
ar.close()
# the product's layout: one 48 GiB step dealt over its teeth; blocks of 1 GiB and 4 GiB
os.environ['BB_ARENA_STEP_GIB'] = '48'
ar = arena.Arena(50 * GIB)
held = []
for size_gib, nfr in ((1, 8000), (4, 32768)):
    rr = []
    for k in range(8):
        t = ar.empty(size_gib * GIB // 4)
        held.append(t)
        rr.append(rate(t[:nfr * bench.SPF], nfr, 29 * k))
    print("48 GiB step, blocks of %d GiB: GB/s %s" % (size_gib, " ".join("%.0f" % x for x in rr)))
    del held[:]
print("probe history of that step:", ar.stats()['probe_history'])
ar.close()
# plain allocations of 4.2 GB for comparison
tt = []
for k in range(8):
    t = torch.empty(32768 * bench.SPF, dtype=torch.float32, device=dev)
    tt.append(rate(t, 32768, 31 * k))
    del t
    torch.cuda.empty_cache()
print("torch.empty 4.2 GB: GB/s", " ".join("%.0f" % x for x in tt))
