#!/usr/bin/env python3
"""The drop-in call of a user of the reference: ``fh.read()`` through the
``baseband.io`` plugin modules returns a NumPy array on the HOST (the decoded output is
16 x the file for 2-bit samples, so this path is bound by the device-to-host copy and by
the host's memory, not by the decode).  File -> HBM -> decode -> pinned -> NumPy, against
the same read kept on the GPU.
usage: python tools/bench_dropin_read.py [MiB of file, default 1024]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import synth                      # noqa: E402
from baseband_amd.plugin import vdif as pv          # noqa: E402

mib = float(sys.argv[1]) if len(sys.argv) > 1 else 1024.
nframes = int(mib * 2 ** 20) // 8032
image, h0 = synth.random_vdif(1, nframes, payload_nbytes=8000, frame_rate=1000)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_dropin.vdif')
image.tofile(path)
nsamp = nframes * 32000


def timed(name, fn, reps=3):
    best = None
    for _ in range(reps):
        with pv.open(path, 'rs', sample_rate=32e6) as fh:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = fn(fh)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        del got
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(case=name, file_MiB=round(nframes * 8032 / 2 ** 20), seconds=round(best, 4),
                          Msamples_per_s=round(nsamp / best / 1e6, 1), out_GBps=round(nsamp * 4 / best / 1e9, 2))),
          flush=True)


timed("fh.read() -> fresh NumPy array (the reference's call)", lambda fh: fh.read())
out = np.empty((nsamp,), np.float32)
out[::1024] = 0                                     # pages touched once
timed("fh.read(out=<NumPy array used before>)", lambda fh: fh.read(out=out))
timed("fh.read_tensor() (result stays on the GPU)", lambda fh: fh.read_tensor())
for chunk in (1 << 24, 1 << 27):
    def loop(fh, chunk=chunk):
        n = 0
        while fh.tell() + chunk <= fh.shape[0]:
            x = fh.read(chunk)
            n += x.shape[0]
        return n
    timed("loop of fh.read(%d) -> NumPy" % chunk, loop)
os.remove(path)

# ---- small reads and a writer, as a script written against the reference does them
image.tofile(path)
for chunk in (32000, 320000, 3200000):
    with pv.open(path, 'rs', sample_rate=32e6) as fh:
        fh.read(chunk)
        n = min(2000, fh.shape[0] // chunk - 1)
        t0 = time.perf_counter()
        for _ in range(n):
            x = fh.read(chunk)
        dt = time.perf_counter() - t0
    print(json.dumps(dict(case="loop of fh.read(%d) -> NumPy" % chunk, reads=n, us_per_read=round(dt / n * 1e6, 1),
                          Msamples_per_s=round(n * chunk / dt / 1e6, 1))), flush=True)
with pv.open(path, 'rs', sample_rate=32e6) as fh:
    h0v = fh.header0
    data = fh.read(64 * 1000 * 32000 // 8)            # 8000 frames = 1.0 GB of float32
os.remove(path)
wpath = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_dropin_w.vdif')
for piece in (data.shape[0], 32000 * 100):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        with pv.open(wpath, 'ws', header0=h0v, sample_rate=32e6, nthread=1) as fw:
            for lo in range(0, data.shape[0], piece):
                fw.write(data[lo:lo + piece])
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(case="fw.write(NumPy), pieces of %d samples" % piece, samples=int(data.shape[0]),
                          seconds=round(best, 4), Msamples_per_s=round(data.shape[0] / best / 1e6, 1),
                          in_GBps=round(data.nbytes / best / 1e9, 2))), flush=True)
os.remove(wpath)
