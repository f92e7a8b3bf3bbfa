"""bench.py's mid-size leg with and without the pre-read of the window, in one process (the leg as
the driver's line reports it: host wall clock between two device synchronisations, median of three)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib                 # noqa: E402
from bench_legs.mid_size import leg_mid_size           # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (8 << 30) // bench.FRAME_NBYTES
image, h0 = bench.make_file_image_on_device(nframes, 4242, 0, dev)
for knob in (256, 0, 256, 0):
    kernels.tune(_lib.TUNE_TOUCH_MIB, knob)
    res = leg_mid_size(dev, image, draws=2, launches=3)
    for r in res['sizes']:
        a = r['api_read']
        print(json.dumps({'pre-read up to MiB': knob, 'frames': r['frames'], 'api_ms': a['ms_median'], 'api_frac': a['frac'],
                          'kernel_ms': a['kernel_ms_same_output'], 'b2b_ms': a['back_to_back']['ms_per_read'],
                          'b2b_frac': a['back_to_back']['frac']}), flush=True)
kernels.tune(_lib.TUNE_TOUCH_MIB, -1)
