#!/usr/bin/env python3
"""Stream writers not covered by bench_writers.py: VDIF 8 threads, GSB."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, gsb   # noqa: E402
from baseband_amd.vdif.header import VDIFHeader   # noqa: E402


def run(case, paths, opener, data, chunk):
    best = None
    for _ in range(3):
        for q in paths:
            if os.path.exists(q):
                os.remove(q)        # (truncating 0.5 GiB of page cache costs 65 ms: not the writer's time)
        torch.cuda.synchronize()
        t = time.perf_counter()
        with opener() as fw:
            for lo in range(0, data.shape[0], chunk):
                fw.write(data[lo:lo + chunk])
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    size = sum(os.path.getsize(p) for p in paths)
    print(json.dumps(dict(case=case, file_GiB=round(size / 2 ** 30, 3), seconds=round(best, 4),
                          file_GBps=round(size / best / 1e9, 2))), flush=True)
    for p in paths:
        os.remove(p)


tmp = os.environ.get('TMPDIR', '/tmp')
g = torch.Generator(device='cuda').manual_seed(1)
t0 = np.datetime64('2014-06-13T05:30:01')
# VDIF 8 threads x 16 ch complex 2-bit: 1000 complex samples per frame
nsets = 8192
data = torch.view_as_complex(torch.randn(nsets * 1000, 8, 16, 2, device='cuda', generator=g) * 2.)
path = os.path.join(tmp, 'bb_w8.vdif')
h0 = VDIFHeader.fromvalues(edv=0, time=t0, nchan=16, bps=2, complex_data=True, thread_id=0,
                           samples_per_frame=1000, station='AA')
run('VDIF 8 threads x 16 ch complex 2-bit', [path],
    lambda: vdif.open(path, 'ws', header0=h0, sample_rate=1e6, nthread=8), data, 1000 * 1024)
del data
# GSB rawdump 4-bit: 64 blocks of 4 MiB
spf = 1 << 23
data = torch.randn(64 * spf, device='cuda', generator=g) * 3.
ts, raw = os.path.join(tmp, 'bb_wr.timestamp'), os.path.join(tmp, 'bb_wr.dat')
run('GSB rawdump 4-bit, 4 MiB blocks', [raw, ts],
    lambda: gsb.open(ts, 'ws', raw=raw, time=t0, samples_per_frame=spf, sample_rate=spf / 0.25165824), data, spf * 8)
del data
# GSB phased 8-bit complex 512 ch, 2 pol x 2 files
spf = (1 << 22) * 2 // (512 * 2)            # samples per frame: two files of 4 MiB per pol
data = torch.view_as_complex(torch.randn(16 * spf, 2, 512, 2, device='cuda', generator=g) * 30.)
ts = os.path.join(tmp, 'bb_wp.timestamp')
raws = tuple(tuple(os.path.join(tmp, 'bb_wp_%d_%d.dat' % (p, f)) for f in range(2)) for p in range(2))
run('GSB phased 8-bit 2 pol x 2 files x 512 ch', [r for pair in raws for r in pair] + [ts],
    lambda: gsb.open(ts, 'ws', raw=raws, time=t0, samples_per_frame=spf, nchan=512,
                     sample_rate=spf / 0.25165824), data, spf * 4)
