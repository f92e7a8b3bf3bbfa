#!/usr/bin/env python3
"""8-bit flat decode: non-temporal loads, kernel variants, grid caps."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
nbytes = 8 << 30
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device='cuda')
out = torch.empty(nbytes, dtype=torch.float32, device='cuda')
payload, header = 128 << 20, 4096
stride = payload + header
nfr = nbytes // stride
alg = nfr * (stride + payload * 4)
def run(tag):
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, payload, _lib.CODER_INT, 8, src0=header,
                                              src_stride=stride, out=out[:nfr * payload]))
    print(json.dumps(dict(tag=tag, ms=round(ms, 3), TBps=round(alg / ms / 1e9, 3))), flush=True)
run('default')
kernels.tune(_lib.TUNE_NT_LOADS, 1); run('nt_loads'); kernels.tune(_lib.TUNE_NT_LOADS, 0)
kernels.tune(_lib.TUNE_NT_STORES, 0); run('plain_stores'); kernels.tune(_lib.TUNE_NT_STORES, 1)
for v in (0, 2, 3, 4, 5):
    kernels.tune(_lib.TUNE_FLAT_VARIANT, v); run('variant%d' % v)
kernels.tune(_lib.TUNE_FLAT_VARIANT, 5)
for b in (2048, 8192, 32768, 131072, 524288):
    kernels.tune(_lib.TUNE_BLOCKS, b); run('blocks%d' % b)
kernels.tune(_lib.TUNE_BLOCKS, 0)
# reference points: copy and fill through torch
x = buf[:nbytes].view(torch.int32)
y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x)); print(json.dumps(dict(tag='torch copy 8+8 GiB', ms=round(ms, 3), TBps=round(2 * nbytes / ms / 1e9, 3))))
ms = timeit(lambda: out.fill_(1.0)); print(json.dumps(dict(tag='torch fill 32 GiB', ms=round(ms, 3), TBps=round(4 * nbytes / ms / 1e9, 3))))
ms = timeit(lambda: torch.sum(x)); print(json.dumps(dict(tag='torch sum 8 GiB (read only)', ms=round(ms, 3), TBps=round(nbytes / ms / 1e9, 3))))
