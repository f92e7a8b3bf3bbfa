"""Is it the WORK ORDER that makes physically contiguous output slow?  2^15 cfg2 frames
(4.2 GB) decoded into (a) blocks of an arena step made of 1 GiB granules (each granule one
physically contiguous GiB: decodes at 5.4 TB/s), (b) blocks of the product's step (32 MiB
granules dealt over 48 teeth: 6.4), (c) plain allocations -- each with the launch dealt
over 1, 2, 4 ... 64 stripes (BB_TUNE_WORK_STRIPES; the product: 4 below 16 GiB of output).
Needs the experiment build.    BB_EXPERIMENTS=1 python tools/experiments/exp_stripes_placement.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GIB = 1 << 30

if len(sys.argv) > 1:
    os.environ['BB_ARENA_STEP_GIB'] = '48'
    os.environ['BB_ARENA_RETRY_BELOW_GBPS'] = '0'
    import torch
    import bench
    from baseband_amd import arena, kernels, _lib
    dev = torch.device('cuda', 0)
    kernels.init()
    nframes = (2 << 30) // bench.FRAME_NBYTES
    image = torch.empty(nframes * bench.FRAME_NBYTES, dtype=torch.uint8, device=dev)
    image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev, into=image)
    nf = 1 << 15

    def rate(out, first):
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

    what = sys.argv[1]
    outs = []
    if what == 'torch':
        for k in range(6):
            outs.append(torch.empty(nf * bench.SPF, dtype=torch.float32, device=dev))
    else:
        ar = arena.Arena(50 * GIB)
        for k in range(6):
            outs.append(ar.empty(nf * bench.SPF))
    for lw in (-1, 0, 1, 2, 3, 4, 5, 6, 8):
        kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
        rr = [rate(o, 31 * k) for k, o in enumerate(outs)]
        print("%-22s stripes %-8s GB/s %s" % (what + ' chunk ' + os.environ.get('BB_ARENA_CHUNK_MIB', '32'),
                                              'product' if lw < 0 else str(1 << lw), " ".join("%.0f" % x for x in rr)), flush=True)
    sys.exit(0)

for chunk, what in (('1024', 'arena'), ('32', 'arena'), ('32', 'torch')):
    env = dict(os.environ, BB_ARENA_CHUNK_MIB=chunk, BB_EXPERIMENTS='1')
    subprocess.run([sys.executable, os.path.abspath(__file__), what], env=env)
