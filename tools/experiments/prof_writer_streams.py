"""cProfile of a VDIF and a Mark 4 stream writer (0.5 GiB files): where the host time goes."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import vdif, mark4
from baseband_amd.vdif.header import VDIFHeader
tmp = os.environ.get('TMPDIR', '/tmp')
g = torch.Generator(device='cuda').manual_seed(1)
t0 = np.datetime64('2014-06-13T05:30:01')
data = torch.randn(32000 * 65536, device='cuda', generator=g) * 2.
h0 = VDIFHeader.fromvalues(edv=0, time=t0, nchan=1, bps=2, complex_data=False, thread_id=0,
                           samples_per_frame=32000, station='AA')


def run(path, opener, d, chunk):
    for rep in range(3):
        if os.path.exists(path):
            os.remove(path)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t = time.perf_counter()
        pr.enable()
        with opener() as fw:
            for lo in range(0, d.shape[0], chunk):
                fw.write(d[lo:lo + chunk])
        pr.disable()
        dt = time.perf_counter() - t
    print(path, round(dt, 4), 's', round(os.path.getsize(path) / dt / 1e9, 2), 'GB/s')
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
    print(s.getvalue()[:3000])
    os.remove(path)


p1 = os.path.join(tmp, 'bb_pw.vdif')
run(p1, lambda: vdif.open(p1, 'ws', header0=h0, sample_rate=32e6, nthread=1), data, 32000 * 8192)
d3 = data[:80000 * 8 * 3000].reshape(-1, 8)
p2 = os.path.join(tmp, 'bb_pw.m4')
run(p2, lambda: mark4.open(p2, 'ws', sample_rate=32e6, ntrack=64, bps=2, fanout=4, time=t0), d3, 80000 * 512)
