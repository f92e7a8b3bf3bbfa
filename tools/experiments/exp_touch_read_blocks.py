"""fh.read() of GUPPI and DADA streams held in HBM with and without the read-only pass queued in front
of the decode (`read_through`), alternating, another block at every read: one and two blocks of 128
MiB (GUPPI 2 pol x 64 channels, channels first; DADA 2 pol complex int8), files written by this
package's writers."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib, guppi, dada          # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nblk = 24


def ab(fh, spf, name):
    for nb in (1, 2):
        count = nb * spf
        nwin = nblk // nb - 1
        ts = {True: [], False: []}
        for r in range(14):
            for on in (True, False):
                fh.read_through = on
                fh.seek(((r * 2 + (not on) + 1) % nwin) * count)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = fh.read(count)
                b.record()
                b.synchronize()
                if r >= 3:
                    ts[on].append(a.elapsed_time(b))
                del out
        w, wo = float(np.median(ts[True])), float(np.median(ts[False]))
        print("%s, %d block(s) per read: with the read-through %.1f us, without %.1f us: x%.3f   [%s]"
              % (name, nb, w * 1e3, wo * 1e3, wo / w, _lib.last_kernel()), flush=True)


tmp = tempfile.mkdtemp()
g = torch.Generator(device=dev)
g.manual_seed(3)
# DADA: 2 pol, 1 channel, complex int8: 128 MiB payloads
spf = (128 << 20) // 4
path = os.path.join(tmp, 'x.dada')
with dada.open(path, 'ws', time=np.datetime64('2013-07-02T01:39:20'), sample_rate=16e6, samples_per_frame=spf, npol=2, nchan=1,
               bps=8, complex_data=True, squeeze=False) as fw:
    for k in range(nblk):
        v = torch.randint(-100, 100, (spf, 2, 1, 2), generator=g, device=dev).to(torch.float32)
        fw.write(torch.view_as_complex(v))
img = torch.from_numpy(np.fromfile(path, np.uint8)).to(dev)
os.remove(path)
with dada.open(img, 'rs', squeeze=False) as fh:
    ab(fh, spf, 'DADA 2 pol complex int8 (128 MiB blocks)')
del img
# GUPPI: 2 pol x 64 channels complex int8, 128 MiB blocks less a little (so that two blocks and their
# headers stay under the limit), both storage orders
T = (120 << 20) // (2 * 64 * 2)
for fmt, name in (('1SFA', 'GUPPI channels first'), ('SIMPLE', 'GUPPI time first')):
    path = os.path.join(tmp, 'x.raw')
    with guppi.open(path, 'ws', time=np.datetime64('2018-01-14T14:11:33'), sample_rate=250e3, samples_per_frame=T, overlap=0,
                    npol=2, nchan=64, bps=8, pktfmt=fmt, squeeze=False) as fw:
        for k in range(nblk):
            v = torch.randint(-100, 100, (T, 2, 64, 2), generator=g, device=dev).to(torch.float32)
            fw.write(torch.view_as_complex(v))
    img = torch.from_numpy(np.fromfile(path, np.uint8)).to(dev)
    os.remove(path)
    with guppi.open(img, 'rs', squeeze=False) as fh:
        ab(fh, T, name + ' (120 MiB blocks)')
    del img
