#!/usr/bin/env python3
"""Thread interleave: k_decode_rows_pipe against k_decode_gather over (thread slots x
floats per thread sample), 8 GiB of 2-bit frames, same tensors per case."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8
bps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
slots = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [2, 4, 8, 16, 64]
chunks = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [32, 64, 128, 256]
nbytes = int(gib * 2 ** 30)
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device=dev)
out = torch.empty(nbytes // 8032 * (64000 // bps), dtype=torch.float32, device=dev)
for nslot in slots:
    nsets = nbytes // 8032 // nslot
    src = (torch.arange(nsets * nslot, device=dev, dtype=torch.int64) * 8032 + 32)
    for chunk in chunks:
        res = {}
        for name, gc in (('rows', 32), ('gather', 4096), ('rows again', 32), ('gather again', 4096)):
            kernels.tune(_lib.TUNE_GATHER_CHUNKS, gc)
            try:
                ms = timeit(lambda: kernels.decode_frames(buf, nsets, 8000, 0, bps, chunk=chunk, nslot=nslot, src=src, out=out), reps=4)
                res[name] = round(nsets * nslot * (8032 + 256000 // bps) / ms / 1e9, 3)
            except Exception as exc:
                res[name] = repr(exc)[:40]
            res[name + ' kernel'] = _lib.last_kernel()[:24]
        kernels.tune(_lib.TUNE_GATHER_CHUNKS, 32)
        print(json.dumps(dict(bps=bps, nslot=nslot, chunk=chunk, TBps=res)), flush=True)
