#!/usr/bin/env python3
"""Does an async H2D copy progress while the host is busy copying memory?"""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

n = 128 << 20
pin = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
dev = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(2)]
src = np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8)
stream = torch.cuda.Stream()
pool = ThreadPoolExecutor(8)


def cpu_copy(dst):
    step = n // 8
    futs = [pool.submit(np.copyto, dst[o:o + step], src[o:o + step]) for o in range(0, n, step)]
    for f in futs:
        f.result()


for p in pin:
    p.fill_(0)
for _ in range(2):
    with torch.cuda.stream(stream):
        dev[0].copy_(pin[0], non_blocking=True)
    stream.synchronize()
for what in ('nothing', 'sleep 4 ms', 'cpu copy into the other pinned buffer', 'cpu copy into pageable memory',
             'busy python loop 4 ms'):
    waits = []
    for rep in range(5):
        with torch.cuda.stream(stream):
            dev[0].copy_(pin[0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        t0 = time.perf_counter()
        if what.startswith('sleep'):
            time.sleep(0.004)
        elif 'other pinned' in what:
            cpu_copy(pin[1].numpy())
        elif 'pageable' in what:
            cpu_copy(np.empty(n, np.uint8))
        elif 'busy' in what:
            while time.perf_counter() - t0 < 0.004:
                pass
        t1 = time.perf_counter()
        ev.synchronize()
        t2 = time.perf_counter()
        waits.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
    print(what, '-> (host work ms, then wait ms):', waits, flush=True)
