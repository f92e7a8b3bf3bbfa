"""Three launches each of the headline decode (cfg2 geometry, 2^20 frames, through
an index) and of the SAME launch with every index entry -1 (stores only), for
`rocprofv3 --pmc <L2 / fabric counters> -- python3 tools/experiments/pmc_rw.py`:
which queue fills when 6 % of the traffic is reads?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from baseband_amd import kernels, _lib          # noqa: E402

kernels.init()
dev = torch.device('cuda', 0)
FB, PB = 8032, 8000
nfr = 1 << 20
img = torch.empty(nfr * FB + 256, dtype=torch.uint8, device=dev)
for lo in range(0, img.numel(), 1 << 30):
    img[lo:lo + (1 << 30)].random_(0, 256)
out = torch.empty(nfr * PB * 4, dtype=torch.float32, device=dev)
src = torch.arange(nfr, dtype=torch.int64, device=dev) * FB + 32
none = torch.full((nfr,), -1, dtype=torch.int64, device=dev)
for which in (src, none, src, none, src, none):
    kernels.decode_frames(img, nfr, PB, _lib.CODER_VDIF, 2, src=which, out=out)
    torch.cuda.synchronize()
print("done")
