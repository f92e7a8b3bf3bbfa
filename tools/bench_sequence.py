#!/usr/bin/env python3
"""End-to-end read of a VDIF stream cut into 16 files at arbitrary byte
positions (helpers.sequentialfile image, windows staged piece by piece)."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth   # noqa: E402

tmp = os.environ.get('TMPDIR', '/tmp')
image, h0 = synth.random_vdif(1, (1 << 30) // 8032, payload_nbytes=8000, frame_rate=1000)
cuts = np.linspace(0, len(image), 17).astype(np.int64)
cuts[1:-1] += 1234                                   # not at frame boundaries
names = []
for i in range(16):
    name = os.path.join(tmp, 'bb_seq_%02d.vdif' % i)
    image[cuts[i]:cuts[i + 1]].tofile(name)
    names.append(name)
single = os.path.join(tmp, 'bb_seq_all.vdif')
image.tofile(single)
del image
for case, src in (('16 files', names), ('one file', single), ('16 files', names), ('one file', single)):
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        with vdif.open(src, 'rs', sample_rate=32e6, verify=False) as fh:
            out = fh.read()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        n = out.shape[0]; del out
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(case=case, seconds=round(best, 4), file_GBps=round((1 << 30) / best / 1e9, 2), samples=n)), flush=True)
for n in names + [single]:
    os.remove(n)
