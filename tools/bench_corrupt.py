#!/usr/bin/env python3
"""Corruption-tolerant path: (1) the byte-granular frame search kernels on an
8 GiB image resident in HBM, (2) open(verify='fix').read() end to end on a
1 GiB VDIF / Mark 5B file with a few damaged places."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, vdif, mark5b, synth   # noqa: E402
from tools.bench_formats import timeit                   # noqa: E402
kernels.init()


def main():
    tmp = os.environ.get('TMPDIR', '/tmp')
    # ---- kernels
    n = (1 << 30) // 8032
    image, h0 = synth.random_vdif(12345, n, payload_nbytes=8000, frame_rate=1000)
    pattern, mask = h0.invariant_pattern()
    dev = torch.from_numpy(np.tile(image, 8)).cuda()
    nb = dev.numel()
    ms = timeit(lambda: kernels.vdif_locate(dev, nb, 8032, 32, pattern, mask), reps=3)
    print(json.dumps(dict(case='bb_vdif_locate, 8 GiB image, 8032-byte frames (incl. sort of the offsets)',
                          ms=round(ms, 2), file_GBps=round(nb / ms / 1e6, 1))), flush=True)
    del dev
    # ---- end to end, VDIF
    path = os.path.join(tmp, 'bb_c.vdif')
    raw = image.copy()
    damaged = np.concatenate([raw[:8032 * 1000], raw[8032 * 1001:8032 * 50000 + 100],     # a frame lost,
                              raw[8032 * 50000 + 300:8032 * 90000],                          # 200 bytes lost,
                              np.full(777, 0x5a, np.uint8), raw[8032 * 90000:]])             # junk inserted
    damaged.tofile(path)
    def rd(verify):
        with vdif.open(path, 'rs', sample_rate=32e6, verify=verify) as fh:
            return fh.read()
    for verify in ('fix',):
        best = None
        import warnings
        for _ in range(3):
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                torch.cuda.synchronize(); t = time.perf_counter()
                out = rd(verify); torch.cuda.synchronize()
                dt = time.perf_counter() - t
            nz = out.shape[0]
            del out
            best = dt if best is None else min(best, dt)
        print(json.dumps(dict(case="VDIF 1 GiB, three damaged places, open(verify='fix').read()",
                              seconds=round(best, 4), file_GBps=round(os.path.getsize(path) / best / 1e9, 2),
                              samples=nz)), flush=True)
    image.tofile(path)
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = rd(True); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        del out
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(case="the intact file, verify=True", seconds=round(best, 4),
                          file_GBps=round(os.path.getsize(path) / best / 1e9, 2))), flush=True)
    os.remove(path)


if __name__ == '__main__':
    main()
