"""Host time per window of a 2 GiB cfg2 / cfg3 VDIF read from the page cache
(cProfile of one read each; profiles/r04zw_prof_window_host.log)."""
import os, sys, time, tempfile, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import baseband_amd as bb
from baseband_amd.vdif.header import VDIFHeader

dev = torch.device('cuda', 0)
tmp = tempfile.mkdtemp(prefix='bb_win_')
g = torch.Generator(device=dev); g.manual_seed(1)
t0 = np.datetime64('2014-06-13T05:30:01')
cases = []
path = os.path.join(tmp, 'cfg2.vdif')
chunk = torch.randn(4096 * 32000, device=dev, generator=g) * 2.
h0 = VDIFHeader.fromvalues(edv=0, time=t0, nchan=1, bps=2, complex_data=False, thread_id=0, samples_per_frame=32000, station='AA')
with bb.vdif.open(path, 'ws', header0=h0, sample_rate=32e6, nthread=1) as fw:
    for _ in range((2 << 30) // 8032 // 4096):
        fw.write(chunk)
cases.append(('cfg2', path, dict(sample_rate=32e6)))
path = os.path.join(tmp, 'cfg3.vdif')
chunk = torch.view_as_complex(torch.randn(1024 * 1000, 8, 16, 2, device=dev, generator=g) * 2.)
h3 = VDIFHeader.fromvalues(edv=0, time=t0, nchan=16, bps=2, complex_data=True, thread_id=0, samples_per_frame=1000, station='AA')
with bb.vdif.open(path, 'ws', header0=h3, sample_rate=1e6, nthread=8) as fw:
    for _ in range((2 << 30) // (8032 * 8) // 1024):
        fw.write(chunk)
cases.append(('cfg3', path, dict(sample_rate=1e6)))
del chunk
for name, path, kw in cases:
    for warm in range(3):
        with bb.vdif.open(path, 'rs', **kw) as fh:
            got = fh.read()
        torch.cuda.synchronize(); del got
    pr = cProfile.Profile()
    fh = bb.vdif.open(path, 'rs', **kw)
    pr.enable()
    got = fh.read()
    pr.disable()
    torch.cuda.synchronize(); fh.close(); del got
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
    print('=====', name); print(s.getvalue()[:3500], flush=True)
    os.remove(path)
