#!/usr/bin/env python3
"""PCIe-inclusive (end-to-end) rate of the drop-in API: open(file).read()
with the file in the page cache: page cache -> pinned -> HBM (side stream)
-> scan/index/decode, double buffered.  Never the headline `value`.

usage: python tools/bench_pipeline.py [GiB, default 2]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, guppi, synth          # noqa: E402
from baseband_amd.guppi.header import GUPPIHeader    # noqa: E402


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    tmp = os.environ.get('TMPDIR', '/tmp')
    # cfg2 layout file
    nframes = int(gib * 2 ** 30) // 8032
    image, h0 = synth.random_vdif(12345, nframes, payload_nbytes=8000, frame_rate=1000)
    path = os.path.join(tmp, 'bb_pipeline.vdif')
    image.tofile(path)
    del image
    for window_mib in (64, 256):
        vdif.VDIFStreamReader.window_bytes = window_mib << 20
        best = None
        for rep in range(3):
            with vdif.open(path, 'rs', sample_rate=32e6, verify=False) as fh:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = fh.read()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                n = out.shape[0]
            del out
            best = dt if best is None else min(best, dt)
        print(json.dumps(dict(case='VDIF cfg2 layout, open().read() end to end (page cache -> HBM -> decode)',
                              file_GiB=round(os.path.getsize(path) / 2 ** 30, 3), window_MiB=window_mib,
                              seconds=round(best, 4), file_GBps=round(os.path.getsize(path) / best / 1e9, 2),
                              Msamples_per_s=round(n / best / 1e6, 1))), flush=True)
    os.remove(path)
    # cfg4a layout file: GUPPI 8-bit, 2 pol, 64 ch, 128 MiB blocks
    npol, nchan, blk = 2, 64, 128 << 20
    nblk = max(1, int(gib * 2 ** 30) // blk)
    h = GUPPIHeader.fromvalues(blocsize=blk, obsnchan=nchan, npol=2 * npol, nbits=8, overlap=0,
                               pktidx=0, pktsize=8192, tbin=1e-6, stt_imjd=58119, stt_smjd=0,
                               stt_offs=0.0)
    import io
    hb = io.BytesIO()
    h.tofile(hb)
    hb = np.frombuffer(hb.getvalue(), np.uint8)
    rng = np.random.default_rng(3)
    path = os.path.join(tmp, 'bb_pipeline.raw')
    with open(path, 'wb') as f:
        for i in range(nblk):
            f.write(hb.tobytes())
            f.write(rng.integers(0, 256, blk, dtype=np.uint8).tobytes())
    best = None
    for rep in range(3):
        with guppi.open(path, 'rs') as fh:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fh.read()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n = out.shape[0]
        del out
        best = dt if best is None else min(best, dt)
    print(json.dumps(dict(case='GUPPI 8-bit 2 pol 64 ch 128 MiB blocks, open().read() end to end',
                          file_GiB=round(os.path.getsize(path) / 2 ** 30, 3),
                          seconds=round(best, 4), file_GBps=round(os.path.getsize(path) / best / 1e9, 2),
                          Msamples_per_s=round(n * npol * nchan / best / 1e6, 1))), flush=True)
    os.remove(path)


if __name__ == '__main__':
    main()
