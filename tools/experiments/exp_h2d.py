#!/usr/bin/env python3
"""Experiment: host -> device copy rate from pinned memory, by chunk size."""
import json
import time

import torch

total = 2 << 30
for mib in (8, 32, 64, 128, 256):
    n = mib << 20
    pinned = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    pinned.fill_(1)
    dev = torch.empty(n, dtype=torch.uint8, device='cuda')
    stream = torch.cuda.Stream()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(total // n):
                dev.copy_(pinned, non_blocking=True)
        stream.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps(dict(chunk_MiB=mib, rep=rep, is_pinned=pinned.is_pinned(),
                              GBps=round(total / dt / 1e9, 2))), flush=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    dev.copy_(pinned, non_blocking=True)
    b.record()
    b.synchronize()
    print(json.dumps(dict(chunk_MiB=mib, single_copy_GBps=round(n / a.elapsed_time(b) / 1e6, 2))), flush=True)
