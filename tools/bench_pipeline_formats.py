#!/usr/bin/env python3
"""End-to-end open().read() (file in the page cache -> HBM -> decode) for the
formats bench_pipeline.py does not cover: VDIF 8 threads, Mark 4, Mark 5B (via
the stream writer), DADA."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, mark4, mark5b, dada, synth   # noqa: E402


def best_of(fn, n=3):
    best = None
    for _ in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        shape = tuple(out.shape)
        del out
        best = dt if best is None else min(best, dt)
    return best, shape


def report(case, path, dt, shape):
    size = os.path.getsize(path)
    print(json.dumps(dict(case=case, file_GiB=round(size / 2 ** 30, 3), seconds=round(dt, 4),
                          file_GBps=round(size / dt / 1e9, 2), shape=shape)), flush=True)


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    tmp = os.environ.get('TMPDIR', '/tmp')
    nbytes = int(gib * 2 ** 30)
    # VDIF 8 threads x 16 channels complex (cfg3 layout)
    path = os.path.join(tmp, 'bb_pf.vdif')
    image, h0 = synth.random_vdif(7, nbytes // (8032 * 8), nthread=8, nchan=16, complex_data=True,
                                  payload_nbytes=8000, frame_rate=1000,
                                  thread_order=[1, 3, 5, 7, 0, 2, 4, 6])
    image.tofile(path); del image
    def r1():
        with vdif.open(path, 'rs', sample_rate=1e6, verify=False) as fh:
            return fh.read()
    dt, shape = best_of(r1)
    report('VDIF 8 threads x 16 ch 2-bit complex', path, dt, shape)
    def r1v():
        with vdif.open(path, 'rs', sample_rate=1e6, verify=True) as fh:
            return fh.read()
    dt, shape = best_of(r1v)
    report('the same, verify=True', path, dt, shape)
    os.remove(path)
    # Mark 4, 64 tracks, fanout 4
    path = os.path.join(tmp, 'bb_pf.m4')
    image, h0 = synth.random_mark4(5, nbytes // 160000, ntrack=64, fanout=4, frame_rate=400)
    image.tofile(path); del image
    def r2():
        with mark4.open(path, 'rs', ntrack=64, decade=2010, sample_rate=32e6, verify=False) as fh:
            return fh.read()
    dt, shape = best_of(r2)
    report('Mark 4 64 tracks fanout 4', path, dt, shape)
    os.remove(path)
    # Mark 5B 16 channels: written with the stream writer (GPU encoder)
    path = os.path.join(tmp, 'bb_pf.m5b')
    nfr = nbytes // 10016
    rng = torch.Generator(device='cuda').manual_seed(4)
    with mark5b.open(path, 'ws', sample_rate=32e6, nchan=16, bps=2,
                     time=np.datetime64('2014-06-13T05:30:01')) as fw:
        for lo in range(0, nfr, 8192):
            n = min(8192, nfr - lo)
            fw.write(torch.randn(n * 2500, 16, device='cuda', generator=rng) * 2.)
    def r3():
        with mark5b.open(path, 'rs', sample_rate=32e6, nchan=16, kday=56000, verify=False) as fh:
            return fh.read()
    dt, shape = best_of(r3)
    report('Mark 5B 16 ch 2-bit', path, dt, shape)
    os.remove(path)
    # DADA 8-bit complex 2 pol, 128 MiB frames
    path = os.path.join(tmp, 'bb_pf.dada')
    from baseband_amd.dada.header import DADAHeader
    spf = (128 << 20) // 4
    h = DADAHeader.fromvalues(time=np.datetime64('2013-07-02T01:39:20'), sample_rate=16e6, bps=8,
                              complex_data=True, npol=2, nchan=1, samples_per_frame=spf)
    nblk = max(1, nbytes // (128 << 20))
    import io
    rg = np.random.default_rng(3)
    with open(path, 'wb') as f:
        for k in range(nblk):
            hk = h.copy()
            hk['OBS_OFFSET'] = k * h.payload_nbytes
            b = io.BytesIO(); hk.tofile(b)
            f.write(b.getvalue())
            f.write(rg.integers(0, 256, h.payload_nbytes, dtype=np.uint8).tobytes())
    def r4():
        with dada.open(path, 'rs') as fh:
            return fh.read()
    dt, shape = best_of(r4)
    report('DADA 8-bit 2 pol complex, 128 MiB frames', path, dt, shape)
    os.remove(path)


if __name__ == '__main__':
    main()
