import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from baseband_amd import vdif, kernels, _lib
from baseband_amd.base import base as bbase
dev = torch.device('cuda', 0)
kernels.init()
nframes = (1 << 30) // bench.FRAME_NBYTES
image, _ = bench.image_buffer(nframes * bench.FRAME_NBYTES, dev)
image, h0 = bench.make_file_image_on_device(nframes, 777, 0, dev, into=image)
SPF = bench.SPF
rng = np.random.default_rng(1)
plan = []
for k in range(12):
    n = int(rng.integers(1 << 9, 1 << 15)) if k % 3 else int(rng.integers(1 << 11, 1 << 15))
    plan.append((int(rng.integers(0, nframes - n)), n, int(rng.integers(0, 2))))
print(plan)
def direct(f0, n):
    return kernels.decode_frames(image, n, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32 + f0 * 8032, src_stride=8032)
for side in (True, False):
    bbase._SIDE_SCAN = side
    fhs = [vdif.open(image, 'rs', sample_rate=float(SPF * bench.FRAME_RATE)) for _ in range(2)]
    for k, (f0, n, which) in enumerate(plan):
        fh = fhs[which]
        fh.seek(f0 * SPF)
        got = fh.read(n * SPF)
        want = direct(f0, n)
        same = torch.equal(got.view(torch.int32), want.view(torch.int32))
        if not same:
            g = got.view(n, SPF); w = want.view(n, SPF)
            badf = torch.nonzero((g.view(torch.int32) != w.view(torch.int32)).any(dim=1)).flatten()
            print("side", side, "read", k, (f0, n, which), "frames that differ:", badf.numel(), badf[:8].tolist(), "values", g[badf[0], :4].tolist(), w[badf[0], :4].tolist())
        del got, want
    for fh in fhs:
        fh.close()
print("done")
