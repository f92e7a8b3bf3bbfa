#!/usr/bin/env python3
"""Kernel-level roofline numbers for the non-headline configurations
(BASELINE.json configs 0, 2, 3, 4): decode kernels on synthetic payload bytes
resident in HBM, timed with HIP events.  Writes one JSON line per case.

usage: python tools/bench_formats.py [GiB of input per case, default 1]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib          # noqa: E402
from baseband_amd.mark4._bitmaps import BITMAPS  # noqa: E402

PEAK = 8000.0


_SWEEP = [int(x) for x in os.environ.get('BB_WORK_STRIPES_SWEEP', '').split(',') if x]
_last_sweep = {}


def _time_once(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def timeit(fn, reps=7):
    """Median launch time (ms).  BB_WORK_STRIPES_SWEEP=2,3,4 (experiment build): the same
    launch -- same buffers, same process, so the same placement -- also under each of these
    work orders (log2 of the number of stripes); `report` attaches the times."""
    global _last_sweep
    _last_sweep = {}
    for lw in _SWEEP:
        kernels.tune(_lib.TUNE_WORK_STRIPES, lw)
        _last_sweep[1 << lw] = _time_once(fn, reps)
    if _SWEEP:
        kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
    return _time_once(fn, reps)


def report(name, ms, bytes_in, bytes_out, nvalues, **extra):
    gbs = (bytes_in + bytes_out) / ms / 1e6
    line = dict(case=name, ms=round(ms, 4), algorithmic_GBps=round(gbs, 1),
                frac_of_8TBps=round(gbs / PEAK, 4),
                Mvalues_per_s=round(nvalues / ms / 1e3, 1),
                bytes_in=bytes_in, bytes_out=bytes_out, **extra)
    if _last_sweep:
        line['GBps_by_stripes'] = {str(k): round((bytes_in + bytes_out) / v / 1e6, 1) for k, v in _last_sweep.items()}
    print(json.dumps(line), flush=True)


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    nbytes = int(gib * 2 ** 30)
    dev = torch.device('cuda')
    kernels.init()
    if os.environ.get('BB_WORK_STRIPES_LW'):            # experiment build: the work order of every launch
        kernels.tune(_lib.TUNE_WORK_STRIPES, int(os.environ['BB_WORK_STRIPES_LW']))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    buf = torch.randint(0, 2 ** 31 - 1, (nbytes // 4 + 1024,), generator=g, device=dev,
                        dtype=torch.int64).to(torch.int32).view(torch.uint8)

    # cfg1 at this size, for comparison with the other cases
    nfr = nbytes // 8032
    out = torch.empty(nfr * 32000, dtype=torch.float32, device=dev)
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, 8000, 0, 2, src0=32, src_stride=8032, out=out))
    report('cfg1 VDIF 1 thread x 1 ch 2-bit real (k_decode_flat_lut)', ms, nfr * 8032,
           out.numel() * 4, out.numel(), nframes=nfr)
    # the same through the dense index, and with 1 % of the frames invalid
    # (index entry -1 -> fill): SURVEY section 8d asks that the fill path cost nothing
    src = torch.arange(nfr, device=dev, dtype=torch.int64) * 8032 + 32
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, 8000, 0, 2, src=src, out=out))
    report('cfg1 through the index, all frames valid', ms, nfr * 8032, out.numel() * 4,
           out.numel(), nframes=nfr)
    bad = torch.rand(nfr, generator=g, device=dev) < 0.01
    src1 = torch.where(bad, torch.full_like(src, -1), src)
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, 8000, 0, 2, src=src1, out=out))
    report('cfg1 through the index, 1 % invalid frames (fill)', ms, nfr * 8032, out.numel() * 4,
           out.numel(), nframes=nfr, invalid=int(bad.sum().item()))
    del out, src, src1

    # cfg0-like: sample.vdif structure, 8 threads x 1 channel, 2-bit real, 5032-byte frames
    fn_, pn, nth = 5032, 5000, 8
    nsets = nbytes // (fn_ * nth)
    src = (torch.arange(nsets * nth, device=dev, dtype=torch.int64) * fn_ + 32)
    out = torch.empty(nsets * nth * pn * 4, dtype=torch.float32, device=dev)
    ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=1, nslot=nth, src=src, out=out))
    report('cfg0 VDIF 8 threads x 1 ch 2-bit real (LDS thread interleave, k_decode_gather)',
           ms, nsets * nth * fn_, out.numel() * 4, out.numel(), nsets=nsets)
    del out

    # cfg2: 8 threads, 16 channels, 2-bit complex, 8032-byte frames (thread order shuffled)
    fn_, pn, nth = 8032, 8000, 8
    nsets = nbytes // (fn_ * nth)
    perm = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=dev)       # slot -> position on disk
    pos = torch.arange(nsets, device=dev, dtype=torch.int64)[:, None] * nth + perm[None, :]
    src = (pos * fn_ + 32).reshape(-1).contiguous()
    out = torch.empty(nsets * nth * pn * 4, dtype=torch.float32, device=dev)
    ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src,
                                              complex_data=True, out=out))
    report('cfg2 VDIF 8 threads x 16 ch 2-bit complex (k_decode_rows_pipe)',
           ms, nsets * nth * fn_, out.numel() * 4, out.numel() // 2, nsets=nsets)
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 2)
    ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=32, nslot=nth, src=src,
                                              complex_data=True, out=out))
    kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
    report('cfg2 same, previous kernel (k_decode_flat_pipe ROWS4, one workgroup per thread)',
           ms, nsets * nth * fn_, out.numel() * 4, out.numel() // 2, nsets=nsets)
    del out

    # cfg3a: Mark 5B 16 ch 2-bit
    nfr = nbytes // 10016
    out = torch.empty(nfr * 40000, dtype=torch.float32, device=dev)
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, 10000, _lib.CODER_MARK5B, 2, chunk=16,
                                              src0=16, src_stride=10016, out=out))
    report('cfg3a Mark5B 16 ch 2-bit (k_decode_flat_pipe)', ms, nfr * 10016, out.numel() * 4,
           out.numel(), nframes=nfr)
    del out

    # cfg3b: Mark 4 64 tracks fanout 4
    m = BITMAPS[(8, 2, 4)]
    nfr = nbytes // 160000
    out = torch.empty(nfr * 20000 * 32, dtype=torch.float32, device=dev)
    ms = timeit(lambda: kernels.decode_mark4(buf, nfr, 64, 20000, m['sign_bit'], m['mag_bit'],
                                             fill_words=160, src0=0, src_stride=160000, out=out))
    report('cfg3b Mark4 64 tracks fanout 4 (k_decode_mark4)', ms, nfr * 160000, out.numel() * 4,
           out.numel(), nframes=nfr)
    del out

    # cfg4a: GUPPI 8-bit, 2 pol complex, 64 channels, channels-first, 128 MiB blocks
    npol, nchan = 2, 64
    blk = 128 << 20
    T = blk // (npol * nchan * 2)
    nfr = max(1, nbytes // blk)
    out = torch.empty(nfr * T * npol * nchan * 2, dtype=torch.float32, device=dev)
    for ov in (0, 512):
        ms = timeit(lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0,
                                                    T - ov, src0=0, src_stride=blk,
                                                    out=out[:nfr * (T - ov) * npol * nchan * 2]))
        nb = nfr * (T - ov) * npol * nchan * 2
        report('cfg4a GUPPI 8-bit 2 pol 64 ch channels-first overlap %d (k_decode_i8_tiled)' % ov,
               ms, nb, nb * 4, nb // 2, nframes=nfr)
    del out

    # cfg4b: DADA 8-bit, 2 pol complex, 1 channel (flat cast)
    nb = nbytes // 4 * 4
    out = torch.empty(nb, dtype=torch.float32, device=dev)
    ms = timeit(lambda: kernels.decode_frames(buf, 1, nb, _lib.CODER_INT, 8, src0=0, out=out))
    report('cfg4b DADA 8-bit 2 pol complex (k_decode_flat_pipe INT8)', ms, nb, nb * 4, nb // 2)
    # write side: 2-bit VDIF encode of 8 GiB worth of payload (read-amplified twin of cfg1)
    del out
    nfl = (nbytes // 8000) * 32000
    x = torch.empty(nfl, dtype=torch.float32, device=dev)
    x.normal_(0., 2.2, generator=g)
    ms = timeit(lambda: kernels.encode_flat(x, 0, 2))
    report('encode: float32 -> 2-bit VDIF codes (k_encode_flat)', ms, nfl * 4, nfl // 4, nfl)
    m = BITMAPS[(8, 2, 4)]
    ms = timeit(lambda: kernels.encode_mark4(x, 64, m['sign_bit'], m['mag_bit']))
    report('encode: float32 -> Mark 4 64-track words (k_encode_mark4)', ms, nfl * 4, nfl // 4, nfl)
    del x
    out = torch.empty(nb, dtype=torch.float32, device=dev)
    # yardstick: the same 1 B -> 4 B cast done by the framework's elementwise kernel
    i8 = buf[:nb].view(torch.int8)
    ms = timeit(lambda: torch.Tensor.copy_(out, i8))
    report('yardstick: torch int8 -> float32 elementwise copy_ (not part of the product)', ms, nb, nb * 4, nb // 2)


if __name__ == '__main__':
    main()
