#!/usr/bin/env python3
"""Frame-at-a-time use of the binary-file API ('rb'): read_frame().data and
Payload.fromfile(...).data in a loop, microseconds per frame."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, mark5b, synth   # noqa: E402
from baseband_amd.vdif.payload import VDIFPayload   # noqa: E402

tmp = os.environ.get('TMPDIR', '/tmp')
path = os.path.join(tmp, 'bb_fl.vdif')
image, h0 = synth.random_vdif(1, 20000, payload_nbytes=8000, frame_rate=1000)
image.tofile(path); del image
n = 5000
for case in ('read_frame().data', 'read_frame() only', 'read_header + VDIFPayload.fromfile(...).data',
             'read_frame().data -> host (numpy)'):
    best = None
    for _ in range(3):
        with vdif.open(path, 'rb') as fh:
            torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(n):
                if case.startswith('read_header'):
                    h = fh.read_header()
                    d = VDIFPayload.fromfile(fh, h).data
                elif case == 'read_frame() only':
                    d = fh.read_frame()
                elif case.endswith('(numpy)'):
                    d = np.asarray(fh.read_frame())
                else:
                    d = fh.read_frame().data
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(case=case, us_per_frame=round(best / n * 1e6, 1),
                          Msamples_per_s=round(n * 32000 / best / 1e6, 1))), flush=True)
os.remove(path)
