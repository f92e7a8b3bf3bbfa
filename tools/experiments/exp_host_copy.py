#!/usr/bin/env python3
"""Experiment: page cache -> pinned buffer rate, mmap copy vs pread, and H2D."""
import json
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * 2 ** 30)
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_hostcopy.bin')
with open(path, 'wb') as f:
    blk = np.random.default_rng(0).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
    for _ in range(n // len(blk)):
        f.write(blk)
win = 128 << 20
pinned = torch.empty(win, dtype=torch.uint8, pin_memory=True)
pn = pinned.numpy()
dev = torch.empty(win, dtype=torch.uint8, device='cuda')


def report(name, dt, nbytes):
    print(json.dumps(dict(case=name, seconds=round(dt, 4), GBps=round(nbytes / dt / 1e9, 2))), flush=True)


for threads in (4, 8, 16, 32):
    pool = ThreadPoolExecutor(threads)
    step = win // threads
    for fresh in (True, False):
        fd = os.open(path, os.O_RDONLY)
        mm = mmap.mmap(fd, n, access=mmap.ACCESS_READ)
        img = np.frombuffer(mm, np.uint8)
        if not fresh:
            for lo in range(0, n, win):
                pn[:] = img[lo:lo + win]            # touch every page once
        t0 = time.perf_counter()
        for lo in range(0, n, win):
            futs = [pool.submit(np.copyto, pn[o:o + step], img[lo + o:lo + o + step])
                    for o in range(0, win, step)]
            for fu in futs:
                fu.result()
        report('mmap copy, %d threads, %s mapping' % (threads, 'fresh' if fresh else 'touched'),
               time.perf_counter() - t0, n)
        del img
        mm.close()
        os.close(fd)
    fd = os.open(path, os.O_RDONLY)
    mv = memoryview(pn)
    t0 = time.perf_counter()
    for lo in range(0, n, win):
        futs = [pool.submit(os.preadv, fd, [mv[o:o + step]], lo + o) for o in range(0, win, step)]
        for fu in futs:
            fu.result()
    report('preadv, %d threads' % threads, time.perf_counter() - t0, n)
    os.close(fd)
    pool.shutdown()
torch.cuda.synchronize()
t0 = time.perf_counter()
for lo in range(0, n, win):
    dev.copy_(pinned, non_blocking=True)
torch.cuda.synchronize()
report('H2D pinned -> device', time.perf_counter() - t0, n)
os.remove(path)
