#!/usr/bin/env python3
"""Frame-at-a-time reading, the way the reference's own loop (and many user
scripts) go through a file: `for ...: fh.read(samples_per_frame)`.
usage: python tools/bench_small_reads.py [MiB, default 256]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth          # noqa: E402

mib = float(sys.argv[1]) if len(sys.argv) > 1 else 256.
nframes = int(mib * 2 ** 20) // 8032
image, h0 = synth.random_vdif(1, nframes, payload_nbytes=8000, frame_rate=1000)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_small.vdif')
image.tofile(path)
for chunk in (32000, 320000, 3200000):
    for verify in (False, 'fix'):
        with vdif.open(path, 'rs', sample_rate=32e6, verify=verify) as fh:
            fh.read(chunk)
            torch.cuda.synchronize()
            n = min(int(os.environ.get('BB_READS', 2000)), fh.shape[0] // chunk - 1)
            t0 = time.perf_counter()
            for _ in range(n):
                x = fh.read(chunk)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(json.dumps(dict(case='read(%d) loop' % chunk, verify=str(verify), reads=n,
                              us_per_read=round(dt / n * 1e6, 1),
                              Msamples_per_s=round(n * chunk / dt / 1e6, 1))), flush=True)
os.remove(path)
