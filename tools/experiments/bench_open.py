#!/usr/bin/env python3
"""Latency of open(name, 'rs') + first small read + info, per format, 1 GiB files."""
import json, os, sys, time, io
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, mark4, mark5b, dada, guppi, synth   # noqa: E402
import baseband_amd                                                  # noqa: E402


def measure(case, opener, nread):
    best = {}
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fh = opener()
        t1 = time.perf_counter()
        fh.read(nread); torch.cuda.synchronize()
        t2 = time.perf_counter()
        fh.stop_time
        t3 = time.perf_counter()
        str(fh.info)
        t4 = time.perf_counter()
        fh.close()
        for k, v in (('open_ms', t1 - t0), ('first_read_ms', t2 - t1), ('stop_time_ms', t3 - t2), ('info_ms', t4 - t3)):
            best[k] = min(best.get(k, 1e9), v * 1e3)
    print(json.dumps(dict(case=case, **{k: round(v, 2) for k, v in best.items()})), flush=True)


def main():
    tmp = os.environ.get('TMPDIR', '/tmp')
    nbytes = 1 << 30
    path = os.path.join(tmp, 'bb_o.vdif')
    image, h0 = synth.random_vdif(1, nbytes // 8032, payload_nbytes=8000, frame_rate=1000)
    image.tofile(path); del image
    measure('VDIF 1 thread, sample_rate given', lambda: vdif.open(path, 'rs', sample_rate=32e6), 32000)
    measure('VDIF 1 thread, frame rate from the file', lambda: vdif.open(path, 'rs'), 32000)
    measure('VDIF 1 thread, baseband_amd.open (format detection)', lambda: baseband_amd.open(path, 'rs'), 32000)
    os.remove(path)
    image, h0 = synth.random_vdif(7, nbytes // (8032 * 8), nthread=8, nchan=16, complex_data=True,
                                  payload_nbytes=8000, frame_rate=1000, thread_order=[1, 3, 5, 7, 0, 2, 4, 6])
    image.tofile(path); del image
    measure('VDIF 8 threads, frame rate from the file', lambda: vdif.open(path, 'rs'), 1000)
    os.remove(path)
    path = os.path.join(tmp, 'bb_o.m4')
    image, h0 = synth.random_mark4(5, nbytes // 160000, ntrack=64, fanout=4, frame_rate=400)
    image.tofile(path); del image
    measure('Mark 4, ntrack and frame rate from the file', lambda: mark4.open(path, 'rs', decade=2010), 80000)
    os.remove(path)
    path = os.path.join(tmp, 'bb_o.m5b')
    with mark5b.open(path, 'ws', sample_rate=32e6, nchan=16, bps=2, time=np.datetime64('2014-06-13T05:30:01')) as fw:
        for lo in range(0, nbytes // 10016, 8192):
            n = min(8192, nbytes // 10016 - lo)
            fw.write(torch.randn(n * 2500, 16, device='cuda') * 2.)
    measure('Mark 5B, frame rate from the file', lambda: mark5b.open(path, 'rs', nchan=16, kday=56000), 2500)
    os.remove(path)
    path = os.path.join(tmp, 'bb_o.raw')
    from baseband_amd.guppi.header import GUPPIHeader
    blk = 16 << 20
    spf = blk // (2 * 64 * 2)
    hg = GUPPIHeader.fromvalues(time=np.datetime64('2014-06-13T05:30:01'), sample_rate=1e6, samples_per_frame=spf,
                                overlap=0, npol=2, nchan=64, pktsize=8192, bps=8)
    rg = np.random.default_rng(3)
    with open(path, 'wb') as f:
        for k in range(nbytes // blk):
            b = io.BytesIO(); hg.tofile(b)
            f.write(b.getvalue()); f.write(rg.integers(0, 256, blk, dtype=np.uint8).tobytes())
    measure('GUPPI 64 blocks of 16 MiB', lambda: guppi.open(path, 'rs'), 1000)
    os.remove(path)


if __name__ == '__main__':
    main()
