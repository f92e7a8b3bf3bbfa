"""How wide must a step be?  Arena steps of 12 .. 192 GiB (each one hipMemCreate batch, chunks
mapped in the product's scattered order), cfg2 decodes of 8000 / 2^15 / 2^17 frames
(1 / 4.2 / 16.8 GB) into eight blocks of each, one process per width.
    python tools/experiments/exp_step_width.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GIB = 1 << 30

if len(sys.argv) > 1:
    width = int(sys.argv[1])
    os.environ['BB_ARENA_STEP_GIB'] = str(width)
    os.environ['BB_ARENA_RETRY_BELOW_GBPS'] = '0'
    import torch
    import bench
    from baseband_amd import arena, kernels, _lib
    dev = torch.device('cuda', 0)
    kernels.init()
    nframes = (4 << 30) // bench.FRAME_NBYTES
    image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
    image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)

    def rate(out, nf, first):
        ts = []
        for r in range(5):
            win = image[((first + r * 7919) % (nframes - nf)) * bench.FRAME_NBYTES:][:nf * bench.FRAME_NBYTES]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            kernels.decode_frames(win, nf, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32, src_stride=bench.FRAME_NBYTES, out=out)
            b.record()
            b.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        return nf * (bench.FRAME_NBYTES + bench.PAYLOAD_NBYTES * 16) / float(np.median(ts)) / 1e6

    ar = arena.Arena((width + 1) * GIB)
    row = []
    for nf in (8000, 1 << 15, 1 << 17):
        if nf * bench.SPF * 4 > width * GIB // 2:
            continue
        held, rr = [], []
        for k in range(4 if nf > (1 << 15) else 8):
            t = ar.empty(nf * bench.SPF)
            if t is None:
                break
            held.append(t)
            rr.append(rate(t, nf, 31 * k))
        del held
        row.append("%6d frames: median %.0f (%.0f-%.0f)" % (nf, np.median(rr), min(rr), max(rr)))
    print("step of %3d GiB, probe %s: %s" % (width, ar.stats()['probe_history'], "  ".join(row)), flush=True)
    sys.exit(0)

for width in (12, 24, 48, 96, 192, 48, 96):
    subprocess.run([sys.executable, os.path.abspath(__file__), str(width)])
