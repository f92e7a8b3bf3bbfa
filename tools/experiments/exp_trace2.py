#!/usr/bin/env python3
"""Per-workgroup timelines from the item completion stamps: duration of the
first item of a workgroup (includes its start-up and exposed first load)
against its later items, for mid-size and large launches."""
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

def run(nfr, blocks):
    kernels.tune(_lib.TUNE_BLOCKS, blocks)
    fn = lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_VDIF, 2, src0=header,
                                       src_stride=stride, out=out[:nfr * per])
    fn(); fn(); torch.cuda.synchronize()
    trace.zero_()
    _lib.lib.bb_debug_trace(C.c_void_p(trace.data_ptr()))
    fn(); torch.cuda.synchronize()
    _lib.lib.bb_debug_trace(None)
    nitems = 2 * nfr
    t = trace[:nitems].cpu().numpy().astype(np.float64)
    t = (t - t.min()) / 100.0                               # microseconds
    G = min(blocks if blocks else 131072, nitems)
    n = nitems // G
    tt = t[:n * G].reshape(n, G)                            # tt[k, b]
    res = dict(frames=nfr, grid=G, items_per_wg=n, total_us=round(float(t.max()), 1))
    if n > 1:
        d = np.diff(tt, axis=0)
        res['later_item_us'] = dict(median=round(float(np.median(d)), 2), p10=round(float(np.percentile(d, 10)), 2),
                                    p90=round(float(np.percentile(d, 90)), 2))
    first = np.sort(tt[0])
    res['first_done_us_quantiles'] = [round(float(np.percentile(first, q)), 1) for q in (0, 1, 10, 50, 90, 100)]
    res['first_round_first_item_us'] = round(float(np.median(tt[0, :2048])), 2)
    if G > 4096:
        order_end = np.sort(tt[-1])[:G - 2048]
        order_first = np.sort(tt[0])[2048:]
        gap = order_first - order_end
        res['successor_first_item_minus_predecessor_end_us'] = dict(
            median=round(float(np.median(gap)), 2), p10=round(float(np.percentile(gap, 10)), 2),
            p90=round(float(np.percentile(gap, 90)), 2))
    print(json.dumps(res), flush=True)

for nfr, blocks in ((1 << 18, 0), (1 << 20, 0), (1 << 18, 32768), (1 << 18, 2048), (1 << 18, 4096), (1 << 20, 2048)):
    run(nfr, blocks)
kernels.tune(_lib.TUNE_BLOCKS, 0)
