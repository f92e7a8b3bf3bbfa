#!/usr/bin/env python3
import cProfile, pstats, io, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth
tmp = os.environ.get('TMPDIR', '/tmp')
image, h0 = synth.random_vdif(1, (1 << 30) // 8032, payload_nbytes=8000, frame_rate=1000)
cuts = np.linspace(0, len(image), 17).astype(np.int64); cuts[1:-1] += 1234
names = []
for i in range(16):
    name = os.path.join(tmp, 'bb_seq_%02d.vdif' % i); image[cuts[i]:cuts[i + 1]].tofile(name); names.append(name)
del image
def go():
    with vdif.open(names, 'rs', sample_rate=32e6, verify=False) as fh:
        out = fh.read()
    torch.cuda.synchronize()
go(); go()
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(30); print(s.getvalue()[:5500])
for n in names: os.remove(n)
