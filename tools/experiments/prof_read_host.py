"""Host time of read() on a resident image, per function (400 reads of 2^12 frames)."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from baseband_amd import vdif, kernels
dev = torch.device('cuda', 0)
kernels.init()
nframes = (1 << 30) // bench.FRAME_NBYTES
image, h0 = bench.make_file_image_on_device(nframes, 12345, 0, dev)
SPF, nf = bench.SPF, 1 << 12
with vdif.open(image, 'rs', sample_rate=float(SPF * bench.FRAME_RATE)) as fh:
    for k in range(20):
        fh.seek(((k * 3 + 1) * nf % (nframes - nf)) * SPF); fh.read(nf * SPF)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for k in range(400):
        fh.seek(((k * 7 + 2) * nf % (nframes - nf)) * SPF)
        got = fh.read(nf * SPF)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
print("per read (under cProfile): %.1f us" % ((t1 - t0) / 400 * 1e6))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
o = s.getvalue(); print(o[o.index('ncalls'):][:4200])
