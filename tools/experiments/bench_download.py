#!/usr/bin/env python3
"""Decoded samples back to the host: tensor.cpu().numpy() against the pinned
double-buffered download (baseband_amd.asnumpy / read(out=ndarray))."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import baseband_amd   # noqa: E402

x = torch.randn(1 << 30, device='cuda')            # 4 GiB of float32
out = np.empty(x.shape, np.float32)
out[:] = 0
for name, fn in (('tensor.cpu().numpy()', lambda: x.cpu().numpy()),
                 ('baseband_amd.asnumpy(tensor, out)', lambda: baseband_amd.asnumpy(x, out))):
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(case=name, GiB=4, seconds=round(best, 3), GBps=round(x.numel() * 4 / best / 1e9, 1))), flush=True)
assert np.array_equal(out[:1000], x[:1000].cpu().numpy()) and np.array_equal(out[-1000:], x[-1000:].cpu().numpy())
z = torch.view_as_complex(torch.randn(1000, 3, 2, device='cuda'))
assert np.array_equal(baseband_amd.asnumpy(z), z.cpu().numpy())
print('ok')
