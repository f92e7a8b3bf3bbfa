#!/usr/bin/env python3
"""Exact workgroup timelines of the aligned flat kernel: start stamp per
workgroup + completion stamp per item -> concurrency, start-to-first-item time,
slot refill behaviour."""
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
    G = min(blocks if blocks else 131072, nitems)
    n = nitems // G
    raw = trace[:nitems + G].cpu().numpy().astype(np.float64)
    t0 = raw[raw > 0].min()
    t = (raw[:nitems] - t0) / 100.0
    start = (raw[nitems:] - t0) / 100.0
    tt = t[:n * G].reshape(n, G)
    end = tt[-1]
    total = float(t.max())
    first_item = tt[0] - start
    life = end - start
    # concurrency: sweep
    ev = np.concatenate([np.stack([start, np.ones(G)], 1), np.stack([end, -np.ones(G)], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0], append=ev[-1, 0])
    avg_conc = float((conc * dt).sum() / total)
    mid = (ev[:, 0] > 0.1 * total) & (ev[:, 0] < 0.9 * total)
    res = dict(frames=nfr, grid=G, items_per_wg=n, total_us=round(total, 1),
               TBps=round(nfr * (stride + payload * 16) / total / 1e6, 2),
               avg_concurrency=round(avg_conc, 1),
               concurrency_mid_p10_p50_p90=[float(np.percentile(conc[mid], q)) for q in (10, 50, 90)] if mid.any() else None,
               start_to_first_item_us=[round(float(np.percentile(first_item, q)), 2) for q in (10, 50, 90)],
               later_item_us=[round(float(np.percentile(np.diff(tt, axis=0), q)), 2) for q in (10, 50, 90)] if n > 1 else None,
               wg_life_us=[round(float(np.percentile(life, q)), 1) for q in (10, 50, 90)])
    # slot refill: time from an end event to the next start event (global, not per CU)
    ends = np.sort(end)
    starts = np.sort(start)
    later_starts = starts[starts > ends[0]]
    k = min(len(later_starts), len(ends))
    refill = later_starts[:k] - ends[:k]
    if k:
        res['kth_start_after_kth_end_us'] = [round(float(np.percentile(refill, q)), 2) for q in (10, 50, 90)]
    res['initially_started'] = int((starts < ends[0]).sum())
    print(json.dumps(res), flush=True)

for nfr, blocks in ((1 << 16, 0), (1 << 17, 0), (1 << 18, 0), (1 << 19, 0), (1 << 20, 0), (1 << 18, 16384), (1 << 18, 2048)):
    run(nfr, blocks)
kernels.tune(_lib.TUNE_BLOCKS, 0)
