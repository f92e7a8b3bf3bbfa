#!/usr/bin/env python3
"""Byte-granular frame search kernels alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, synth
kernels.init()
n = (1 << 30) // 8032
image, h0 = synth.random_vdif(12345, n, payload_nbytes=8000, frame_rate=1000)
pattern, mask = h0.invariant_pattern()
dev = torch.from_numpy(np.tile(image, 8)).cuda()
nb = dev.numel()
for _ in range(4):
    offs = kernels.vdif_locate(dev, nb, 8032, 32, pattern, mask)
torch.cuda.synchronize()
print(offs.numel(), nb)
for _ in range(4):
    offs = kernels.mark5b_locate(dev, nb)
    offs = kernels.mark4_locate(dev, nb, 64)
torch.cuda.synchronize()
