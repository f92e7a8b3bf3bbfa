#!/usr/bin/env python3
"""cProfile of sequential read(65536) on GUPPI 128 MiB blocks."""
import cProfile, io, os, pstats, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import guppi
from baseband_amd.guppi.header import GUPPIHeader
tmp = os.environ.get('TMPDIR', '/tmp')
path = os.path.join(tmp, 'bb_bs_prof.raw')
blk = 128 << 20
spf = blk // (2 * 64 * 2)
hg = GUPPIHeader.fromvalues(time=np.datetime64('2014-06-13T05:30:01'), sample_rate=1e6, samples_per_frame=spf,
                            overlap=0, npol=2, nchan=64, pktsize=8192, bps=8)
rg = np.random.default_rng(3)
with open(path, 'wb') as f:
    for k in range(6):
        b = io.BytesIO(); hg.tofile(b)
        f.write(b.getvalue()); f.write(rg.integers(0, 256, blk, dtype=np.uint8).tobytes())
n = 65536
with guppi.open(path, 'rs') as fh:
    for _ in range(10):
        fh.read(n)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t = time.perf_counter()
    pr.enable()
    reps = 32
    for _ in range(reps):
        fh.read(n)
    torch.cuda.synchronize()
    pr.disable()
    print('us per read', (time.perf_counter() - t) / reps * 1e6)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
print('\n'.join(s.getvalue().splitlines()[:34]))
os.remove(path)
