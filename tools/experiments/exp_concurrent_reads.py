#!/usr/bin/env python3
"""Would a streaming prefetcher make the input reads cheaper?  The headline decode
with its input served by the caches (sources in a 131 MB window) runs next to a
separate streaming read of the whole 8.6 GB input on a second stream; the pair's
elapsed time against the ordinary launch (which does the same reads itself, 256
bytes at a time, from 8000 waves)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
FN, PN = 8032, 8000
nfr = (8 << 30) // FN
buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
out = torch.empty(nfr * PN * 4, dtype=torch.float32, device=dev)
src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32
w128 = (torch.arange(nfr, device=dev, dtype=torch.int64) % 16384) * FN + 32
words = buf[:(nfr * FN) // 8 * 8].view(torch.int64)
side = torch.cuda.Stream()
res = {}
res['decode, real index'] = round(timeit(lambda: kernels.decode_frames(buf, nfr, PN, 0, 2, src=src, out=out), reps=6), 3)
res['decode, cached window'] = round(timeit(lambda: kernels.decode_frames(buf, nfr, PN, 0, 2, src=w128, out=out), reps=6), 3)
res['streaming read alone (sum)'] = round(timeit(lambda: words.sum(), reps=6), 3)


def pair(nparts):
    # the read is cut into `nparts` launches spread over the side stream
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    n = words.numel()
    with torch.cuda.stream(side):
        for k in range(nparts):
            words[k * n // nparts:(k + 1) * n // nparts].sum()
    kernels.decode_frames(buf, nfr, PN, 0, 2, src=w128, out=out)
    main.wait_stream(side)


for nparts in (1, 8, 64):
    res['cached decode || streaming read in %d launches' % nparts] = round(timeit(lambda: pair(nparts), reps=6), 3)
res['decode, real index again'] = round(timeit(lambda: kernels.decode_frames(buf, nfr, PN, 0, 2, src=src, out=out), reps=6), 3)
print(json.dumps(res, indent=1))
