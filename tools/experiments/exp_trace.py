#!/usr/bin/env python3
"""Throughput over time INSIDE one launch: the flat kernels stamp the device
wall clock when a work item is done (bb_debug_trace); items done per 0.25 ms
bin -> TB/s."""
import ctypes as C, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
kernels.init()
payload, header = 8000, 32
stride = payload + header
nmax = 1 << 20
buf = torch.randint(0, 256, (nmax * stride + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nmax * payload * 4, dtype=torch.float32, device='cuda')
trace = torch.zeros(4 * nmax, dtype=torch.int64, device='cuda')
per = payload * 4

def run(nfr, variant, blocks=0):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, variant)
    kernels.tune(_lib.TUNE_BLOCKS, blocks)
    fn = lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                       src_stride=stride, out=out[:nfr * per])
    fn(); fn(); torch.cuda.synchronize()
    trace.zero_()
    _lib.lib.bb_debug_trace(C.c_void_p(trace.data_ptr()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    _lib.lib.bb_debug_trace(None)
    t = trace.cpu().numpy()
    t = t[t > 0]
    nitems = len(t)
    item_bytes = nfr * (stride + payload * 16) / nitems
    t = (t - t.min()) / 100e6 * 1e3                     # ms since the first completion
    bins = np.arange(0, t.max() + 0.25, 0.25)
    hist, _ = np.histogram(t, bins)
    rate = hist * item_bytes / 0.25e-3 / 1e12
    print(json.dumps(dict(frames=nfr, variant=variant, blocks=blocks, items=nitems, event_ms=round(a.elapsed_time(b), 3),
                          span_ms=round(float(t.max()), 3),
                          TBps_per_quarter_ms=[round(float(r), 2) for r in rate])), flush=True)

for nfr in (1 << 17, 1 << 18, 1 << 20):
    for variant in (5, 0):
        run(nfr, variant)
run(1 << 17, 5, 8192)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5); kernels.tune(_lib.TUNE_BLOCKS, 0)
