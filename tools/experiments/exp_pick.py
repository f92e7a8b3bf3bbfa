"""A/B of k_decode_pick against k_decode_gather_select at the bench's shape
(8 GiB of 8-thread 16-channel 2-bit complex VDIF, 2 of 16 channels kept; also 1
and 4 of 16, and 8 threads x 1 channel real keeping all: rowlen 8) -- same
process, same buffers, bit-identical outputs; ms by HIP events, median of 7."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from baseband_amd import kernels, _lib                  # noqa: E402
import baseband_amd                                     # noqa: E402

dev = torch.device('cuda')
kernels.init()
gib = 8.0
nth, nch, pn, fn_ = 8, 16, 8000, 8032
nsets = int(gib * 2 ** 30) // (fn_ * nth)
g = torch.Generator(device=dev); g.manual_seed(1)
buf = torch.empty(nsets * nth * fn_ + 4096, dtype=torch.uint8, device=dev)
for lo in range(0, buf.numel() // 4, 1 << 28):
    hi = min(buf.numel() // 4, lo + (1 << 28))
    buf.view(torch.int32)[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
src = (torch.arange(nsets * nth, device=dev, dtype=torch.int64) * fn_ + 32).contiguous()
spf = pn * 4 // (2 * nch)


def run(within, label):
    w = torch.tensor(within, dtype=torch.int32, device=dev)
    n = nsets * spf * nth * len(within)
    out = baseband_amd.empty_output((n,), dtype=torch.float32, device=dev)
    ref = None
    rows = []
    for name, pick, pb in (("gather_select", 0, 8192), ("pick 8 KiB", 2, 8192), ("pick 4 KiB", 2, 4096), ("pick 2 KiB", 2, 2048),
                           ("pick 16 KiB", 2, 16384)):
        kernels.tune(_lib.TUNE_SELECT_PICK, pick)
        kernels.tune(_lib.TUNE_PICK_BYTES, pb)
        fn = lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=2 * nch, nslot=nth, src=src,
                                           complex_data=True, out=out, within=w)
        med, mean = bench.timed_launches(fn, 7)
        moved = nsets * nth * fn_ + n * 4
        if ref is None:
            ref = out.clone()
            same = True
        else:
            same = bool(torch.equal(ref.view(torch.int32), out.view(torch.int32)))
        rows.append((name, med, moved / med / 1e6, same, _lib.last_kernel()))
    print("## " + label + " (bytes moved {:.2f} GB)".format(moved / 1e9))
    for name, med, gbs, same, k in rows:
        print("  {:14s} {:7.3f} ms  {:7.1f} GB/s  {:.4f} of 8 TB/s  identical {}   {}".format(name, med, gbs, gbs / 8000, same, k))
    del out, ref
    kernels.tune(_lib.TUNE_SELECT_PICK, 1); kernels.tune(_lib.TUNE_PICK_BYTES, 4096)


run([6, 7, 24, 25], "2 of 16 complex channels (bench row)")
run([6, 7], "1 of 16 complex channels")
run([0, 1, 6, 7, 24, 25, 30, 31], "4 of 16 complex channels")
run(list(range(16)), "8 of 16 complex channels")

# the whole thread sample kept (32 of 32 = the cfg3 decode itself): k_decode_pick against k_decode_gather
w = torch.arange(32, dtype=torch.int32, device=dev)
n = nsets * spf * nth * 32
out = baseband_amd.empty_output((n,), dtype=torch.float32, device=dev) if n * 4 <= (64 << 30) else torch.empty(n, dtype=torch.float32, device=dev)
moved = nsets * nth * fn_ + n * 4
rows = []
fn = lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=2 * nch, nslot=nth, src=src, complex_data=True, out=out)
med, _ = bench.timed_launches(fn, 7)
ref = out[::257].clone()
rows.append(("k_decode_gather (product)", med, _lib.last_kernel(), True))
for pb in (8192, 16384, 32768):
    kernels.tune(_lib.TUNE_SELECT_PICK, 2)
    kernels.tune(_lib.TUNE_PICK_BYTES, pb)
    fn = lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=2 * nch, nslot=nth, src=src, complex_data=True, out=out, within=w)
    med, _ = bench.timed_launches(fn, 7)
    rows.append(("pick %d KiB" % (pb >> 10), med, _lib.last_kernel(), bool(torch.equal(ref.view(torch.int32), out[::257].view(torch.int32)))))
kernels.tune(_lib.TUNE_SELECT_PICK, 1); kernels.tune(_lib.TUNE_PICK_BYTES, 4096)
print("## all 16 channels (the cfg3 decode; bytes moved {:.2f} GB)".format(moved / 1e9))
for name, med, k, same in rows:
    print("  {:26s} {:7.3f} ms  {:7.1f} GB/s  {:.4f} of 8 TB/s  identical {}   {}".format(name, med, moved / med / 1e6, moved / med / 1e6 / 8000, same, k))
