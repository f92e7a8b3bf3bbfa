"""Does handing the file mapping to the background reaper at close()
(`staging.retire_image`) pay in a loop over 2 GiB files?  Whole cycles timed,
nothing left outside the clock (cfg2 VDIF; profiles/r04zy_prof_close.log)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import baseband_amd as bb
from baseband_amd.vdif.header import VDIFHeader

dev = torch.device('cuda', 0)
tmp = tempfile.mkdtemp(prefix='bb_close_')
path = os.path.join(tmp, 'cfg2.vdif')
g = torch.Generator(device=dev); g.manual_seed(1)
chunk = torch.randn(4096 * 32000, device=dev, generator=g) * 2.
h0 = VDIFHeader.fromvalues(edv=0, time=np.datetime64('2014-06-13T05:30:01'), nchan=1, bps=2, complex_data=False,
                           thread_id=0, samples_per_frame=32000, station='AA')
with bb.vdif.open(path, 'ws', header0=h0, sample_rate=32e6, nthread=1) as fw:
    for _ in range((2 << 30) // 8032 // 4096):
        fw.write(chunk)
del chunk
with open(path, 'rb') as f:
    while f.read(64 << 20):
        pass
from baseband_amd import staging
got = None
print('whole loops of 10 x (open, read, sync, close, del), ms per cycle; alternating retire off/on')
for rep in range(4):
    for mode in (False, True):
        staging._RETIRE = mode
        for warm in range(2):
            with bb.vdif.open(path, 'rs', sample_rate=32e6) as fh:
                got = fh.read()
            torch.cuda.synchronize()
            del got
        time.sleep(0.1)
        t = time.perf_counter()
        for r in range(10):
            fh = bb.vdif.open(path, 'rs', sample_rate=32e6)
            got = fh.read()
            torch.cuda.synchronize()
            fh.close()
            del got, fh
        dt = (time.perf_counter() - t) / 10
        print('retire', int(mode), 'cycle %.2f ms = %.2f GB/s' % (dt * 1e3, os.path.getsize(path) / dt / 1e9), flush=True)
os.remove(path)
