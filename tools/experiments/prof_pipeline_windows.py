#!/usr/bin/env python3
"""Where does the time of a windowed VDIF read go?  (VDIF
end to end 42-49 GB/s against 52-55 GB/s for pinned H2D alone.)

A cfg2-layout file in the page cache is read with ``open().read()`` while
`staging.WindowPipeline.run` is replaced by an instrumented copy of itself:
per window the host time of the page-cache -> pinned copy, the host time spent
waiting for a pinned buffer to come free, the H2D time and the scan / index /
decode time by events.  Beside it: the pinned H2D rate alone and the
multi-threaded page-cache copy alone, same window size.
    python tools/prof_pipeline_windows.py [GiB, default 2] [window MiB, default 64]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth, staging          # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
win_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tmp = os.environ.get('TMPDIR', '/tmp')
nframes = int(gib * 2 ** 30) // 8032
image, h0 = synth.random_vdif(12345, nframes, payload_nbytes=8000, frame_rate=1000)
path = os.path.join(tmp, 'bb_prof_pipeline.vdif')
image.tofile(path)
fsize = image.size
del image
rows = []


def run(self, ranges, process, sink=None):
    if self._copy_stream is None:
        self._copy_stream = torch.cuda.Stream(device=self.device)
    main = torch.cuda.current_stream(self.device)
    for i, (lo, hi) in enumerate(ranges):
        n = hi - lo
        b = self._count % self.nbuf
        self._count += 1
        t0 = time.perf_counter()
        if self._done[b] is not None:
            self._done[b].synchronize()
        t1 = time.perf_counter()
        pinned, dev = self._buffers(b, need_dev=(sink is None) or OLD[0])
        target = dev[:n] if sink is None else sink[lo:hi]
        t2 = time.perf_counter()
        staging._stage(pinned.numpy(), self.image, lo, hi)
        t3 = time.perf_counter()
        with torch.cuda.stream(self._copy_stream):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(self._copy_stream)
            target.copy_(pinned[:n], non_blocking=True)
            copied = torch.cuda.Event(enable_timing=True)
            copied.record(self._copy_stream)
        main.wait_event(copied)
        k0 = torch.cuda.Event(enable_timing=True)
        k0.record(main)
        process(target, i)
        done = torch.cuda.Event(enable_timing=True)
        done.record(main)
        t4 = time.perf_counter()
        self._done[b] = done
        rows.append(dict(bytes=n, wait_ms=(t1 - t0) * 1e3, buffers_ms=(t2 - t1) * 1e3, stage_ms=(t3 - t2) * 1e3,
                         enqueue_ms=(t4 - t3) * 1e3, ev=(e0, copied, k0, done)))


staging.WindowPipeline.run = run
vdif.VDIFStreamReader.window_bytes = win_mib << 20
OLD = [False]
KEEP = staging._PINNED_KEEP
# A/B in one process: round 2's pipeline (rotating device buffers allocated although the windows go
# to the sink, pinned buffers allocated per reader, all windows full size) against this round's,
# the latter with two and with three pinned buffers
for rep in range(12):
    OLD[0] = rep % 3 == 1
    staging._NBUF = 3 if rep % 3 == 2 else 2
    staging._PINNED_KEEP = 0 if OLD[0] else KEEP
    if OLD[0]:
        staging.release_pinned()
    vdif.VDIFStreamReader.ramp_windows = not OLD[0]
    rows.clear()
    t_open = time.perf_counter()
    with vdif.open(path, 'rs', sample_rate=32e6, verify=False) as fh:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fh.read()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    open_ms = (t0 - t_open) * 1e3
    del out
    h2d = [r['ev'][0].elapsed_time(r['ev'][1]) for r in rows]
    ker = [r['ev'][2].elapsed_time(r['ev'][3]) for r in rows]
    tot = {k: sum(r[k] for r in rows) for k in ('wait_ms', 'buffers_ms', 'stage_ms', 'enqueue_ms')}
    nb = sum(r['bytes'] for r in rows)
    print(json.dumps(dict(rep=rep, pipeline="round 2" if OLD[0] else "round 3, {} buffers".format(staging._NBUF), file_GiB=round(fsize / 2 ** 30, 3), window_MiB=win_mib, windows=len(rows),
                          read_s=round(dt, 4), file_GBps=round(fsize / dt / 1e9, 2), open_ms=round(open_ms, 1),
                          host_ms={k: round(v, 1) for k, v in tot.items()},
                          host_ms_unaccounted=round(dt * 1e3 - sum(tot.values()), 1),
                          stage_GBps=round(nb / tot['stage_ms'] / 1e6, 1),
                          h2d_ms_sum=round(sum(h2d), 1), h2d_GBps=round(nb / sum(h2d) / 1e6, 1),
                          kernels_ms_sum=round(sum(ker), 2),
                          first_windows=[dict(wait=round(r['wait_ms'], 2), stage=round(r['stage_ms'], 2),
                                              enqueue=round(r['enqueue_ms'], 2), h2d=round(h, 2), kernels=round(k, 3))
                                         for r, h, k in list(zip(rows, h2d, ker))[:4]])), flush=True)

# the two stages alone
img = np.memmap(path, dtype=np.uint8, mode='r')
w = win_mib << 20
pin = [torch.empty(w, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
dev = torch.empty(w + 256, dtype=torch.uint8, device='cuda')
nwin = fsize // w
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(nwin):
        staging._stage(pin[i % 2].numpy(), img, i * w, (i + 1) * w)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(case="page cache -> pinned alone ({} threads)".format(staging._COPY_THREADS),
                          GBps=round(nwin * w / dt / 1e9, 1))), flush=True)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(nwin):
        dev[:w].copy_(pin[i % 2], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(case="pinned -> HBM alone", GBps=round(nwin * w / dt / 1e9, 1))), flush=True)
for threads in (4, 8, 16, 32):
    staging._COPY_THREADS = threads
    staging._copy_pool = None
    t0 = time.perf_counter()
    for i in range(nwin):
        staging._stage(pin[i % 2].numpy(), img, i * w, (i + 1) * w)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(case="page cache -> pinned alone", threads=threads, GBps=round(nwin * w / dt / 1e9, 1))), flush=True)
os.remove(path)
