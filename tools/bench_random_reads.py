#!/usr/bin/env python3
"""Random access: seek to a random frame, read one frame's worth of samples."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, synth   # noqa: E402

tmp = os.environ.get('TMPDIR', '/tmp')
path = os.path.join(tmp, 'bb_rr.vdif')
image, h0 = synth.random_vdif(1, (1 << 30) // 8032, payload_nbytes=8000, frame_rate=1000)
image.tofile(path); del image
rng = np.random.default_rng(0)
for verify in (False, True):
    with vdif.open(path, 'rs', sample_rate=32e6, verify=verify) as fh:
        nfr = fh.shape[0] // 32000
        fh.read(32000); torch.cuda.synchronize()
        where = rng.integers(0, nfr - 2, 300)
        t = time.perf_counter()
        for k in where:
            fh.seek(int(k) * 32000 + 137)
            d = fh.read(32000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(json.dumps(dict(case='random seek + read(32000)', verify=str(verify),
                              us_per_read=round(dt / len(where) * 1e6, 1))), flush=True)
        fh.seek(0)
        t = time.perf_counter()
        for k in range(300):
            d = fh.read(32000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(json.dumps(dict(case='sequential read(32000)', verify=str(verify),
                              us_per_read=round(dt / 300 * 1e6, 1))), flush=True)
os.remove(path)
