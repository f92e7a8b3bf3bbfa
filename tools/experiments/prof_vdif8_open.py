#!/usr/bin/env python3
"""Where does open().read() of an 8-thread VDIF file spend its host time?"""
import cProfile, pstats, os, sys, time, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth
tmp = os.environ.get('TMPDIR', '/tmp')
path = os.path.join(tmp, 'bb_prof.vdif')
image, h0 = synth.random_vdif(7, (1 << 30) // (8032 * 8), nthread=8, nchan=16, complex_data=True,
                              payload_nbytes=8000, frame_rate=1000, thread_order=[1, 3, 5, 7, 0, 2, 4, 6])
image.tofile(path); del image
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fh = vdif.open(path, 'rs', sample_rate=1e6, verify=False)
    t1 = time.perf_counter()
    out = fh.read(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    fh.close(); del out
    print('open %.1f ms, read %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
fh = vdif.open(path, 'rs', sample_rate=1e6, verify=False)
out = fh.read(); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
os.remove(path)
