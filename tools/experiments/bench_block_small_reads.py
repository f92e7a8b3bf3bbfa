#!/usr/bin/env python3
"""Loops of small reads on block formats (GUPPI 128 MiB blocks)."""
import json, os, sys, time, io
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import guppi   # noqa: E402
from baseband_amd.guppi.header import GUPPIHeader   # noqa: E402

tmp = os.environ.get('TMPDIR', '/tmp')
path = os.path.join(tmp, 'bb_bs.raw')
blk = 128 << 20
spf = blk // (2 * 64 * 2)
hg = GUPPIHeader.fromvalues(time=np.datetime64('2014-06-13T05:30:01'), sample_rate=1e6, samples_per_frame=spf,
                            overlap=0, npol=2, nchan=64, pktsize=8192, bps=8)
rg = np.random.default_rng(3)
with open(path, 'wb') as f:
    for k in range(4):
        b = io.BytesIO(); hg.tofile(b)
        f.write(b.getvalue()); f.write(rg.integers(0, 256, blk, dtype=np.uint8).tobytes())
with guppi.open(path, 'rs') as fh:
    for n in (1024, 65536):
        fh.seek(0)
        fh.read(n); fh.read(n); torch.cuda.synchronize()     # second touch stages the whole block
        reps = min(300, (fh.shape[0] - n) // n - 1)
        t = time.perf_counter()
        for _ in range(reps):
            d = fh.read(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(json.dumps(dict(case='GUPPI sequential read(%d), 128 MiB blocks' % n,
                              us_per_read=round(dt / reps * 1e6, 1))), flush=True)
    # the same loop with GPU work on every chunk (what a pipeline does between
    # reads): with the next block prefetched in the background the staging of
    # a block hides behind that work
    work = torch.randn(4096, 4096, device='cuda')
    for _ in range(3):
        work @ work                             # (library start-up outside the timing)
    torch.cuda.synchronize()
    for prefetch in (False, True, False, True):
        fh.prefetch_next = prefetch
        fh.seek(0)
        fh.read(n); fh.read(n); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            d = fh.read(n)
            for _ in range(2):
                work @ work                     # ~0.4 ms of matrix work per chunk
            torch.cuda.synchronize()            # (a consumer that needs each result)
        dt = time.perf_counter() - t
        print(json.dumps(dict(case='GUPPI sequential read(%d) + GPU work per chunk' % n,
                              prefetch_next_block=prefetch, us_per_iteration=round(dt / reps * 1e6, 1))), flush=True)
    fh.prefetch_next = True
    rng = np.random.default_rng(1)
    where = rng.integers(0, fh.shape[0] - 2048, 200)
    fh.seek(int(where[0])); fh.read(1024); torch.cuda.synchronize()
    t = time.perf_counter()
    for k in where:
        fh.seek(int(k)); d = fh.read(1024)
    torch.cuda.synchronize()
    print(json.dumps(dict(case='GUPPI random seek + read(1024), 128 MiB blocks',
                          us_per_read=round((time.perf_counter() - t) / len(where) * 1e6, 1))), flush=True)
    # across a block boundary
    fh.seek(spf - 500)
    a = fh.read(1000)
    fh.seek(spf - 500)
    whole = fh.read(spf)          # pipeline path
    print(json.dumps(dict(case='boundary read equals the large read', ok=bool((a == whole[:1000]).all()))))
os.remove(path)
