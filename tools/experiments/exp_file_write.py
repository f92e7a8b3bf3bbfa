"""Where a stream writer's time goes: how fast can this host put 1 GiB that
lies in a pinned buffer into a NEW file (page cache), by write(), by
os.pwrite from several threads, through a memory map filled by several
threads; and rewriting a file that exists (pages already allocated).
    python tools/experiments/exp_file_write.py
"""
import json
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

n = 1 << 30
buf_t = torch.empty(n, dtype=torch.uint8, pin_memory=True)
buf = buf_t.numpy()
buf[:] = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_wtest.bin')


def t_write(fresh=True):
    if fresh and os.path.exists(path):
        os.remove(path)
    t = time.perf_counter()
    with open(path, 'r+b' if not fresh else 'wb', buffering=0) as f:
        f.write(memoryview(buf))
    return time.perf_counter() - t


def t_pwrite(k, fresh=True):
    if fresh and os.path.exists(path):
        os.remove(path)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT)
    t = time.perf_counter()
    step = n // k
    with ThreadPoolExecutor(k) as ex:
        list(ex.map(lambda i: os.pwrite(fd, memoryview(buf[i * step:(i + 1) * step]), i * step), range(k)))
    dt = time.perf_counter() - t
    os.close(fd)
    return dt


def t_mmap(k, fresh=True):
    if fresh and os.path.exists(path):
        os.remove(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT)
    t = time.perf_counter()
    os.ftruncate(fd, n)
    m = mmap.mmap(fd, n)
    a = np.frombuffer(m, dtype=np.uint8)
    step = n // k

    def cp(i):
        a[i * step:(i + 1) * step] = buf[i * step:(i + 1) * step]
    with ThreadPoolExecutor(k) as ex:
        list(ex.map(cp, range(k)))
    dt = time.perf_counter() - t
    del a
    m.close()
    os.close(fd)
    return dt


def t_remove():
    t = time.perf_counter()
    os.remove(path)
    return time.perf_counter() - t


cases = [('write, new file', lambda: t_write()), ('pwrite x4, new file', lambda: t_pwrite(4)),
         ('pwrite x8, new file', lambda: t_pwrite(8)), ('pwrite x16, new file', lambda: t_pwrite(16)),
         ('mmap x1, new file', lambda: t_mmap(1)), ('mmap x8, new file', lambda: t_mmap(8)),
         ('mmap x16, new file', lambda: t_mmap(16)),
         ('write, existing file', lambda: t_write(False)), ('pwrite x8, existing file', lambda: t_pwrite(8, False)),
         ('mmap x8, existing file', lambda: t_mmap(8, False))]
for name, fn in cases:
    ts = [fn() for _ in range(3)]
    print(json.dumps({"case": name, "GBps": round(n / min(ts) / 1e9, 2), "all_s": [round(t, 3) for t in ts]}), flush=True)
print(json.dumps({"case": "remove 1 GiB file", "s": round(t_remove(), 4)}))
if torch.cuda.is_available():
    dev = torch.empty(n, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        buf_t.copy_(dev)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    print(json.dumps({"case": "D2H 1 GiB into the pinned buffer", "GBps": round(n / min(ts) / 1e9, 2)}))
