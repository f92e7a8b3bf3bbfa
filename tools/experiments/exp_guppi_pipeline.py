#!/usr/bin/env python3
"""Where does the GUPPI end-to-end read spend its time?"""
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import guppi, staging, kernels     # noqa: E402
from baseband_amd.guppi.header import GUPPIHeader    # noqa: E402

npol, nchan, blk = 2, 64, 128 << 20
nblk = 16
h = GUPPIHeader.fromvalues(blocsize=blk, obsnchan=nchan, npol=2 * npol, nbits=8, overlap=0,
                           pktidx=0, pktsize=8192, tbin=1e-6, stt_imjd=58119, stt_smjd=0, stt_offs=0.0)
hb = io.BytesIO()
h.tofile(hb)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_g.raw')
rng = np.random.default_rng(3)
with open(path, 'wb') as f:
    for i in range(nblk):
        f.write(hb.getvalue())
        f.write(rng.integers(0, 256, blk, dtype=np.uint8).tobytes())

acc = {}


def timed(mod, name):
    fn = getattr(mod, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.) + time.perf_counter() - t0
        return r
    setattr(mod, name, wrap)


timed(staging, '_stage')
timed(kernels, 'decode_i8_tiled')
# finer: time the pieces of WindowPipeline.run
import types
def run(self, ranges, process):
    if self._copy_stream is None:
        self._copy_stream = torch.cuda.Stream(device=self.device)
    main = torch.cuda.current_stream(self.device)
    T = acc
    for i, (lo, hi) in enumerate(ranges):
        n = hi - lo
        b = i % self.nbuf
        t = time.perf_counter()
        if self._done[b] is not None:
            self._done[b].synchronize()
        T.setdefault('waits', []).append(round((time.perf_counter() - t) * 1e3, 2))
        T['wait'] = T.get('wait', 0) + time.perf_counter() - t; t = time.perf_counter()
        pinned, dev = self._buffers(b)
        T['buffers'] = T.get('buffers', 0) + time.perf_counter() - t; t = time.perf_counter()
        staging._stage(pinned.numpy(), self.image, lo, hi)
        t = time.perf_counter()
        with torch.cuda.stream(self._copy_stream):
            dev[:n].copy_(pinned[:n], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self._copy_stream)
        if os.environ.get('BB_QUERY'):
            self._copy_stream.query()
        T['enqueue_h2d'] = T.get('enqueue_h2d', 0) + time.perf_counter() - t; t = time.perf_counter()
        main.wait_event(copied)
        process(dev[:n], i)
        done = torch.cuda.Event()
        done.record(main)
        self._done[b] = done
        if os.environ.get('BB_QUERY') == '2':
            main.query()
        T['process'] = T.get('process', 0) + time.perf_counter() - t
staging.WindowPipeline.run = run
staging._COPY_THREADS = int(os.environ.get('BB_COPY_THREADS', staging._COPY_THREADS))
_init = staging.WindowPipeline.__init__
def init(self, image, cap, nbuf=2, device='cuda'):
    _init(self, image, cap, nbuf=int(os.environ.get('BB_NBUF', 2)), device=device)
staging.WindowPipeline.__init__ = init
if os.environ.get('BB_WIN'):
    guppi.GUPPIStreamReader.window_bytes = int(os.environ['BB_WIN']) << 20
for rep in range(3):
    acc.clear()
    t_open = time.perf_counter()
    with guppi.open(path, 'rs') as fh:
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fh.read()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print('rep', rep, 'open %.4f read-call %.4f total %.4f' % (t1 - t_open, t2 - t0, dt),
          {k: (round(v, 4) if not isinstance(v, list) else v) for k, v in acc.items()}, flush=True)
    del out
os.remove(path)
