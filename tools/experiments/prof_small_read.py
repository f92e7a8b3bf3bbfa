#!/usr/bin/env python3
import cProfile, pstats, io, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth
tmp = os.environ.get('TMPDIR', '/tmp'); path = os.path.join(tmp, 'bb_ps.vdif')
image, h0 = synth.random_vdif(1, 40000, payload_nbytes=8000, frame_rate=1000); image.tofile(path); del image
with vdif.open(path, 'rs', sample_rate=32e6, verify=False) as fh:
    for _ in range(200): fh.read(32000)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): fh.read(32000)
    torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
os.remove(path)
