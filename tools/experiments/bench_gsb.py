#!/usr/bin/env python3
"""End-to-end GSB reads (timestamp file + raw files in the page cache -> HBM
-> decode): rawdump 4-bit and phased 8-bit (2 pols x 2 files)."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import gsb            # noqa: E402
from baseband_amd.gsb.header import GSBHeader   # noqa: E402


def timestamps(path, mode, nframes, t0):
    h = GSBHeader.fromvalues(mode, time=t0, **({'seq_nr': 0, 'mem_block': 0} if mode == 'phased' else {}))
    with open(path, 'w') as f:
        for k in range(nframes):
            hk = h.copy()
            hk.update(time=t0 + np.timedelta64(int(round(k * 0.25165824e9)), 'ns'),
                      **({'seq_nr': k, 'mem_block': k % 8} if mode == 'phased' else {}))
            f.write(' '.join(hk.words) + '\n')


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


def main():
    tmp = os.environ.get('TMPDIR', '/tmp')
    t0 = np.datetime64('2015-06-01T01:02:03')
    rng = np.random.default_rng(5)
    pn = 1 << 22
    # rawdump: 256 blocks of 4 MiB = 1 GiB
    nfr = 256
    ts, raw = os.path.join(tmp, 'bb_r.timestamp'), os.path.join(tmp, 'bb_r.dat')
    timestamps(ts, 'rawdump', nfr, t0)
    rng.integers(0, 256, nfr * pn, dtype=np.uint8).tofile(raw)
    def rd():
        with gsb.open(ts, 'rs', raw=raw) as fh:
            return fh.read()
    dt, shape = best_of(rd)
    print(json.dumps(dict(case='GSB rawdump 4-bit, 1 GiB', seconds=round(dt, 4), file_GBps=round(nfr * pn / dt / 1e9, 2),
                          shape=shape)), flush=True)
    os.remove(raw); os.remove(ts)
    # phased: 2 pols x 2 files x 64 blocks of 4 MiB = 1 GiB
    nfr = 64
    ts = os.path.join(tmp, 'bb_p.timestamp')
    timestamps(ts, 'phased', nfr, t0)
    raws = tuple(tuple(os.path.join(tmp, 'bb_p_%d_%d.dat' % (p, f)) for f in range(2)) for p in range(2))
    for pair in raws:
        for r in pair:
            rng.integers(0, 256, nfr * pn, dtype=np.uint8).tofile(r)
    def rp():
        with gsb.open(ts, 'rs', raw=raws) as fh:
            return fh.read()
    dt, shape = best_of(rp)
    print(json.dumps(dict(case='GSB phased 8-bit 2 pol x 2 files x 512 ch, 1 GiB', seconds=round(dt, 4),
                          file_GBps=round(4 * nfr * pn / dt / 1e9, 2), shape=shape)), flush=True)
    for pair in raws:
        for r in pair:
            os.remove(r)
    os.remove(ts)


if __name__ == '__main__':
    main()
