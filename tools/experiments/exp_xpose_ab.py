"""A/B of two builds of the library on the byte-format transposes (GUPPI channels-first /
time-first, MKBF heaps), 31 GiB of int8 in -> 124 GiB of float32 out, the bench's shapes:
    python tools/experiments/exp_xpose_ab.py [path/to/libbbdecode.so]   (one build per process)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import baseband_amd._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
from baseband_amd import kernels, _lib
from bench_legs.common import timed_launches
kernels.init()
dev = torch.device('cuda', 0)
nbytes = 31 << 30
buf = torch.empty(nbytes + 4096, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev); g.manual_seed(4242)
for lo in range(0, buf.numel() // 4, 1 << 28):
    hi = min(buf.numel() // 4, lo + (1 << 28))
    buf.view(torch.int32)[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
out = torch.empty(nbytes, dtype=torch.float32, device=dev)
npol, nchan, blk = 2, 64, 128 << 20
T = blk // (npol * nchan * 2)
nfr = nbytes // blk
rows = []
def add(name, fn, nb):
    med, mean = timed_launches(fn, 7)
    rows.append((name, round(mean, 3), round(med, 3), round(5 * nb / mean / 1e6 / 8000, 4), _lib.last_kernel()))
for rep in range(2):
    add("guppi_cf", lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=out[:nfr * blk]), nfr * blk)
    add("guppi_tf", lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npol, nchan, T, 0, T, src0=0, src_stride=blk, out=out[:nfr * blk]), nfr * blk)
    Tm = 256 * 64
    blkm = Tm * npol * nchan * 2
    nfm = nbytes // blkm
    add("mkbf", lambda: kernels.decode_i8_tiled(buf, nfm, _lib.LAYOUT_MKBF, npol, nchan, Tm, 0, Tm, src0=0, src_stride=blkm, out=out[:nfm * blkm]), nfm * blkm)
    # a channel list (mapped channels): 32 of 64
    cm = torch.arange(0, 64, 2, device=dev, dtype=torch.int32)
    try:
        add("guppi_cf_32of64", lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_CF, npol, 32, T, 0, T, src0=0, src_stride=blk, out=out[:nfr * blk // 2], nchan_stored=64, chan_map=cm), nfr * blk // 2)
    except (TypeError, KeyError) as exc:
        if rep == 0: print("no cmap argument:", exc)
print(json.dumps({"lib": L.LIB_PATH, "rows": rows}))
