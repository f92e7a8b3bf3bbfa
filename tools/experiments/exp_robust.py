#!/usr/bin/env python3
"""Which launch geometry is least sensitive to where an output lies?  Fresh
2^17-frame cfg2 tensors are allocated until one decodes slowly (< 5.7 TB/s) and
one quickly (> 6.2) with the default kernel; both are kept and every knob
setting is timed on both."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import kernels, _lib
from tools.bench_formats import timeit
kernels.init()
dev = torch.device('cuda')
FN, PN, SPF = 8032, 8000, 32000
nfr = 1 << 17
buf = torch.randint(0, 256, (nfr * FN + 4096,), dtype=torch.uint8, device=dev)
src = torch.arange(nfr, device=dev, dtype=torch.int64) * FN + 32


def rate(out, reps=5):
    ms = timeit(lambda: kernels.decode_frames(buf, nfr, PN, _lib.CODER_VDIF, 2, src=src, out=out), reps=reps)
    return round(nfr * (FN + SPF * 4) / ms / 1e9, 3)


held, slow, fast = [], None, None
for k in range(14):
    out = torch.empty(nfr * SPF, dtype=torch.float32, device=dev)
    r = rate(out, 3)
    print(json.dumps(dict(candidate=k, TBps=r, ptr=hex(out.data_ptr()))), flush=True)
    if r < 5.7 and slow is None:
        slow = out
    elif r > 6.2 and fast is None:
        fast = out
    held.append(out)            # keep it, so that the next one lands elsewhere
    if slow is not None and fast is not None:
        break
if slow is None or fast is None:
    print(json.dumps(dict(note="no slow/fast pair found in this process")))
    sys.exit(0)
held = [t for t in held if t is slow or t is fast]
torch.cuda.empty_cache()
settings = [('default', [])]
for tpw in (2, 4, 8, 16):
    settings.append(('tiles/wave %d' % tpw, [(_lib.TUNE_TILES_PER_WAVE, tpw)]))
for blocks in (8192, 32768, 524288, 1 << 22):
    settings.append(('grid cap %d' % blocks, [(_lib.TUNE_BLOCKS, blocks)]))
for lw in (0, 2, 4, 6, 8):
    settings.append(('stripes 2^%d' % lw, [(_lib.TUNE_WORK_STRIPES, lw)]))
settings.append(('register select kernel', [(_lib.TUNE_BYTE_LUT, 0)]))
settings.append(('plain kernel (variant 0)', [(_lib.TUNE_FLAT_VARIANT, 0)]))
settings.append(('tiles/wave 4 + grid cap 2^22', [(_lib.TUNE_TILES_PER_WAVE, 4), (_lib.TUNE_BLOCKS, 1 << 22)]))
settings.append(('tiles/wave 2 + grid cap 2^22', [(_lib.TUNE_TILES_PER_WAVE, 2), (_lib.TUNE_BLOCKS, 1 << 22)]))
settings.append(('plain stores (no nt)', [(_lib.TUNE_NT_STORES, 0)]))
defaults = {_lib.TUNE_TILES_PER_WAVE: 12, _lib.TUNE_BLOCKS: 0, _lib.TUNE_WORK_STRIPES: -1,
            _lib.TUNE_BYTE_LUT: 1, _lib.TUNE_FLAT_VARIANT: 5, _lib.TUNE_NT_STORES: 1}
for name, knobs in settings:
    for k, v in knobs:
        kernels.tune(k, v)
    rs, rf = rate(slow), rate(fast)
    kern = _lib.last_kernel()
    for k, _ in knobs:
        kernels.tune(k, defaults[k])
    print(json.dumps(dict(setting=name, slow_TBps=rs, fast_TBps=rf, kernel=kern[:70])), flush=True)
