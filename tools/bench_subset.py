#!/usr/bin/env python3
"""Channel subsets: decode with the selection folded into the kernel
(bb_decode_frames_select) against decode-everything-then-index, on 4 GiB of
Mark 5B 16-channel and 8-thread x 16-channel complex VDIF input in HBM."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
nbytes = 4 << 30
buf = torch.randint(0, 256, (nbytes + 4096,), dtype=torch.uint8, device=dev)
from baseband_amd import arena
ar = arena.Arena(120 << 30)
out = ar.empty(17 << 30)                  # (arena memory: what the readers allocate from)
# Mark 5B 16 channels 2 bit
nfr = nbytes // 10016
src = torch.arange(nfr, device=dev, dtype=torch.int64) * 10016 + 16
full = lambda: kernels.decode_frames(buf, nfr, 10000, _lib.CODER_MARK5B, 2, chunk=16, src=src, out=out[:nfr * 40000])
ms_full = timeit(full, reps=3)
for sel in ([1, 6], [0, 1, 2, 3, 4, 5, 6, 7], list(range(16))):
    w = torch.tensor(sel, dtype=torch.int32, device=dev)
    n = nfr * 2500 * len(sel)
    ms_sel = timeit(lambda: kernels.decode_frames(buf, nfr, 10000, _lib.CODER_MARK5B, 2, chunk=16, src=src,
                                                  out=out[:n], within=w), reps=3)
    idx = torch.tensor(sel, device=dev)
    def two_pass():
        d = full().view(-1, 16)
        return d[:, idx].contiguous()
    ms_two = timeit(two_pass, reps=3)
    print(json.dumps(dict(case="Mark 5B 16 ch 2-bit, %d of 16 channels" % len(sel), kernel=_lib.last_kernel().split(' grid')[0],
                          folded_ms=round(ms_sel, 3), decode_then_index_ms=round(ms_two, 3), full_decode_ms=round(ms_full, 3),
                          folded_GBps_moved=round((nfr * 10016 + n * 4) / ms_sel / 1e6, 1))), flush=True)
# VDIF 8 threads x 16 channels complex
fn, pn, nth = 8032, 8000, 8
nsets = nbytes // (fn * nth)
src8 = (torch.arange(nsets * nth, device=dev, dtype=torch.int64) * fn + 32)
full8 = lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src8, complex_data=True,
                                      out=out[:nsets * nth * pn * 4])
ms_full = timeit(full8, reps=3)
for chans in ([3], [3, 4, 5, 6], list(range(16))):
    sel = [2 * c + k for c in chans for k in (0, 1)]
    w = torch.tensor(sel, dtype=torch.int32, device=dev)
    n = nsets * 1000 * nth * len(sel)
    ms_sel = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src8, complex_data=True,
                                                  out=out[:n], within=w), reps=3)
    idx = torch.tensor(sel, device=dev)
    def two_pass8():
        d = full8().view(-1, nth, 32)
        return d[:, :, idx].contiguous()
    ms_two = timeit(two_pass8, reps=3)
    print(json.dumps(dict(case="VDIF 8 thr x 16 ch complex, %d of 16 channels" % len(chans),
                          folded_ms=round(ms_sel, 3), decode_then_index_ms=round(ms_two, 3), full_decode_ms=round(ms_full, 3),
                          folded_GBps_moved=round((nsets * nth * fn + n * 4) / ms_sel / 1e6, 1))), flush=True)
# Mark 4, 64 tracks fanout 4 (8 channels): shorter bit maps (bb_decode_mark4_select)
from baseband_amd.mark4._bitmaps import BITMAPS
maps = BITMAPS[(8, 2, 4)]
nf4 = nbytes // 160000
full4 = lambda: kernels.decode_mark4(buf, nf4, 64, 20000, maps['sign_bit'], maps['mag_bit'], fill_words=160,
                                     src0=0, src_stride=160000, out=out[:nf4 * 640000])
ms_full = timeit(full4, reps=3)
for chans in ([5], [0, 5, 7], [0, 1, 2, 3], list(range(8))):
    sign, mag = kernels.mark4_select_maps(maps['sign_bit'], maps['mag_bit'], 8, chans)
    n = nf4 * 20000 * len(sign)
    ms_sel = timeit(lambda: kernels.decode_mark4(buf, nf4, 64, 20000, sign, mag, fill_words=160, src0=0,
                                                 src_stride=160000, out=out[:n], select=True), reps=3)
    kname = _lib.last_kernel().split(' grid')[0]
    idx = torch.tensor(chans, device=dev)
    def two_pass4():
        d = full4().view(-1, 8)
        return d[:, idx].contiguous()
    ms_two = timeit(two_pass4, reps=3)
    print(json.dumps(dict(case="Mark 4 64 tracks fanout 4, %d of 8 channels" % len(chans),
                          kernel=kname,
                          folded_ms=round(ms_sel, 3), decode_then_index_ms=round(ms_two, 3), full_decode_ms=round(ms_full, 3),
                          folded_GBps_moved=round((nf4 * 160000 + n * 4) / ms_sel / 1e6, 1))), flush=True)
