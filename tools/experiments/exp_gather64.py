#!/usr/bin/env python3
"""Thread interleave with many thread slots and narrow chunks (LDS gather):
16 / 32 / 64 slots x chunks of 1..16 floats on 8 GiB of 2-bit input (the
headline's output size), staging 8 / 16 / 32 KiB of payload per work item."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
nbytes = int(gib * 2 ** 30)
buf = torch.randint(0, 256, (nbytes + 8192,), dtype=torch.uint8, device=dev)
pn, fn = 8000, 8032
out = torch.empty(nbytes // fn * pn * 4, dtype=torch.float32, device=dev)
for nth in (16, 32, 64):
    nsets = nbytes // (fn * nth)
    src = (torch.arange(nsets * nth, device=dev, dtype=torch.int64) * fn + 32)
    for chunk in (1, 2, 4, 16):
        row = {"threads": nth, "chunk": chunk}
        for gb in (8192, 16384, 32768, 65536):
            kernels.tune(_lib.TUNE_GATHER_BYTES, gb if gb != 8192 else 8191)
            try:
                ms = timeit(lambda: kernels.decode_frames(buf, nsets, pn, 0, 2, chunk=chunk, nslot=nth, src=src,
                                                          complex_data=chunk % 2 == 0, out=out[:nsets * nth * pn * 4]), reps=3)
                row["stage_%dKiB" % (gb // 1024)] = round(nsets * nth * (fn + pn * 16) / ms / 1e9, 3)
            except Exception as exc:
                row["stage_%dKiB" % (gb // 1024)] = repr(exc)[:60]
        row["kernel"] = _lib.last_kernel()
        kernels.tune(_lib.TUNE_GATHER_BYTES, 8192)
        print(json.dumps(row), flush=True)
