#!/usr/bin/env python3
"""Pin windows of a read-only file mapping in place (bb_host_register) and DMA
them to HBM, against the staged upload (host threads copy into pinned buffers).
2 GiB file in the page cache; bytes compared."""
import json, mmap, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from baseband_amd import _lib, kernels
from baseband_amd.staging import upload
kernels.init()
lib = _lib.lib
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_reg.bin')
n = 2 << 30
rng = np.random.default_rng(1)
with open(path, 'wb') as f:
    for _ in range(n >> 26):
        f.write(rng.integers(0, 256, 1 << 26, dtype=np.uint8).tobytes())
fd = os.open(path, os.O_RDONLY)
mm = mmap.mmap(fd, n, access=mmap.ACCESS_READ)
img = np.frombuffer(mm, dtype=np.uint8)
base = img.ctypes.data
dev = torch.empty(n, dtype=torch.uint8, device='cuda')
stream = torch.cuda.Stream()
for chunk in (16 << 20, 64 << 20, 256 << 20, n):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        treg = 0.0
        spans = [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]
        events = []
        for lo, hi in spans:
            t1 = time.perf_counter()
            _lib.check(lib.bb_host_register(base + lo, hi - lo), 'bb_host_register')
            treg += time.perf_counter() - t1
            _lib.check(lib.bb_copy_to_device(dev.data_ptr() + lo, base + lo, hi - lo, stream.cuda_stream), 'bb_copy_to_device')
            ev = torch.cuda.Event(); ev.record(stream); events.append(ev)
        stream.synchronize()
        t2 = time.perf_counter()
        for lo, hi in spans:
            _lib.check(lib.bb_host_unregister(base + lo), 'bb_host_unregister')
        t3 = time.perf_counter()
        print(json.dumps(dict(chunk_MiB=chunk >> 20, rep=rep, total_ms=round((t2 - t0) * 1e3, 2), register_ms=round(treg * 1e3, 2),
                              unregister_ms=round((t3 - t2) * 1e3, 2), GBps=round(n / (t2 - t0) / 1e9, 1))), flush=True)
ok = bool((dev[:1 << 28].cpu().numpy() == img[:1 << 28]).all()) and bool((dev[-(1 << 26):].cpu().numpy() == img[-(1 << 26):]).all())
print(json.dumps(dict(bytes_equal=ok)))
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); d = upload(img); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(json.dumps(dict(staged_upload_ms=round(dt * 1e3, 2), GBps=round(n / dt / 1e9, 1))), flush=True)
os.remove(path)
