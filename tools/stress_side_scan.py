"""Soak of the side-stream scan (kernels._FrameWindow: four sets of scratch taking
turns): 1,500 reads of random sizes and places from a resident 1 GiB image, no host
syncs in between, results dropped at once -- the checksums of every read must equal
those of the same sequence with BB_SIDE_SCAN off; also interleaved on two readers."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from baseband_amd import vdif, kernels                  # noqa: E402
from baseband_amd.base import base as bbase             # noqa: E402

dev = torch.device('cuda', 0)
kernels.init()
nframes = (1 << 30) // bench.FRAME_NBYTES
image, _ = bench.image_buffer(nframes * bench.FRAME_NBYTES, dev)
image, h0 = bench.make_file_image_on_device(nframes, 777, 0, dev, into=image)
SPF = bench.SPF
rng = np.random.default_rng(1)
plan = []
for k in range(1500):
    n = int(rng.integers(1 << 9, 1 << 15)) if k % 3 else int(rng.integers(1 << 11, 1 << 15))
    plan.append((int(rng.integers(0, nframes - n)), n, int(rng.integers(0, 2))))


def run(side):
    bbase._SIDE_SCAN = side
    sums = torch.zeros(len(plan), dtype=torch.float64, device=dev)
    picks = torch.zeros(len(plan), 8, dtype=torch.float32, device=dev)
    fhs = [vdif.open(image, 'rs', sample_rate=float(SPF * bench.FRAME_RATE)) for _ in range(2)]
    for k, (f0, n, which) in enumerate(plan):
        fh = fhs[which]
        fh.seek(f0 * SPF)
        got = fh.read(n * SPF)
        sums[k] = got.double().sum()
        picks[k] = got[::max(1, got.numel() // 8)][:8]
        del got
    torch.cuda.synchronize()
    used = [fh._scan_stream is not None for fh in fhs]
    for fh in fhs:
        fh.close()
    return sums.cpu().numpy(), picks.cpu().numpy(), used


def reference():
    """the same checksums from direct kernel launches, a sync after each"""
    from baseband_amd import _lib
    sums = np.zeros(len(plan))
    picks = np.zeros((len(plan), 8), np.float32)
    for k, (f0, n, which) in enumerate(plan):
        got = kernels.decode_frames(image, n, bench.PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=32 + f0 * 8032, src_stride=8032)
        sums[k] = float(got.double().sum())
        picks[k] = got[::max(1, got.numel() // 8)][:8].cpu().numpy()
        del got
        torch.cuda.synchronize()
    return sums, picks


ref = reference()
a = run(True)
a2 = run(True)
b = run(False)
for name, r in (("side stream", a), ("side stream, again", a2), ("caller's stream", b)):
    wrong = np.nonzero((r[0] != ref[0]) | (r[1] != ref[1]).any(axis=1))[0]
    print("%-20s %d reads differ from the direct decode %s" % (name, len(wrong), [(int(k), plan[k]) for k in wrong[:5]]))
bad = np.nonzero((a[0] != b[0]) | (a[1] != b[1]).any(axis=1))[0]
big = sum(1 for f0, n, w in plan if n * bench.FRAME_NBYTES >= (16 << 20))
for k in bad[:10]:
    print("  differs: read", k, plan[k], "previous", plan[k - 1] if k else None, a[0][k], b[0][k])
print("reads %d (%d of them on the side stream), side stream used %s / %s: %d reads differ%s"
      % (len(plan), big, a[2], b[2], len(bad), '' if not len(bad) else ' -- FIRST ' + str(plan[bad[0]])))
sys.exit(1 if len(bad) else 0)
