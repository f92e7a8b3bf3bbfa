"""The pre-read of the window (exp_touch_read.py) for Mark 5B and Mark 4: files written by this
package's writers from random samples, held in HBM, fh.read() of 63 / 126 / 251 MiB windows,
alternating with and without, another window at every read."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib, mark5b, mark4          # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
levels = torch.tensor([-3.316505, -1.0, 1.0, 3.316505], device=dev)


def image_of(mod, nframes, spf, width, **kw):
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    path = os.path.join(tempfile.mkdtemp(), 'image.bin')
    with mod.open(path, 'ws', squeeze=False, **kw) as fw:
        for lo in range(0, nframes, 2048):
            n = min(2048, nframes - lo)
            fw.write(levels[torch.randint(0, 4, (n * spf, width), generator=g, device=dev)])
    image = torch.from_numpy(np.fromfile(path, np.uint8)).to(dev)
    os.remove(path)
    return image


def ab(fh, frame_nbytes, spf, nframes_file, name):
    for mib in (63, 126, 251):
        nf = (mib << 20) // frame_nbytes
        count = nf * spf
        nwin = nframes_file // nf - 1
        ts = {256: [], 0: []}
        for r in range(14):
            for knob in (256, 0):
                kernels.tune(_lib.TUNE_TOUCH_MIB, knob)
                fh.seek(((r * 2 + (knob == 0) + 1) % nwin) * count)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = fh.read(count)
                b.record()
                b.synchronize()
                if r >= 3:
                    ts[knob].append(a.elapsed_time(b))
                del out
        on, off = float(np.median(ts[256])), float(np.median(ts[0]))
        print("%s, %d MiB of file per read: with pre-read %.1f us, without %.1f us: x%.3f   [%s]"
              % (name, mib, on * 1e3, off * 1e3, off / on, _lib.last_kernel()), flush=True)
    kernels.tune(_lib.TUNE_TOUCH_MIB, -1)


n5 = (1 << 30) // 10016
img = image_of(mark5b, n5, 5000, 8, sample_rate=32e6, nchan=8, bps=2, time=np.datetime64('2014-06-13T05:30:01'))
with mark5b.open(img, 'rs', sample_rate=32e6, nchan=8, bps=2, kday=56000) as fh:
    ab(fh, 10016, 5000, n5, 'Mark 5B, 8 channels 2-bit')
del img
n4 = (1 << 30) // 160000
img = image_of(mark4, n4, 80000, 8, sample_rate=32e6, ntrack=64, fanout=4, nchan=8, bps=2,
               time=np.datetime64('2014-06-16T07:38:12.475'))
with mark4.open(img, 'rs', sample_rate=32e6, ntrack=64, decade=2010) as fh:
    ab(fh, 160000, 80000, n4, 'Mark 4, 64 tracks fan-out 4')
