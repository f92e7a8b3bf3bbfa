#!/usr/bin/env python3
"""cProfile of the frame-at-a-time loop (host overhead of one small read)."""
import cProfile, io, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth
nframes = (64 << 20) // 8032
image, h0 = synth.random_vdif(1, nframes, payload_nbytes=8000, frame_rate=1000)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_small_prof.vdif')
image.tofile(path)
for verify in (False, 'fix'):
    with vdif.open(path, 'rs', sample_rate=32e6, verify=verify) as fh:
        for _ in range(50):
            fh.read(32000)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(2000):
            fh.read(32000)
        torch.cuda.synchronize()
        pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
    print('verify =', verify)
    print('\n'.join(s.getvalue().splitlines()[:40]))
os.remove(path)
