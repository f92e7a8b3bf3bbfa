"""Follow-up of exp_file_write.py (writers at 7-11 GB/s of
file bytes against 52 for reads).  One new file takes 12 GB/s through the page
cache whatever the number of writing threads (profiles/r03y_exp_file_write.log:
buffered writes to one file serialise on its inode lock).  Here: do SEVERAL
files written at the same time scale (a sequence writer could fill the files of
a ``{file_nr}`` template in parallel), what does O_DIRECT give, what a tmpfs
file (/dev/shm), and where is /tmp?
    python tools/experiments/exp_file_write2.py
"""
import json
import mmap
import os
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

n = 1 << 30
buf_t = torch.empty(n, dtype=torch.uint8, pin_memory=True)
buf = buf_t.numpy()
buf[:] = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
tmp = os.environ.get('TMPDIR', '/tmp')
print(json.dumps({"df": subprocess.run(['df', '-T', tmp, '/dev/shm'], capture_output=True, text=True).stdout}))


def many_files(root, k, direct=False):
    """k files of n / k bytes each, one thread per file, write() of 16 MiB pieces"""
    paths = [os.path.join(root, 'bb_w2_{}.bin'.format(i)) for i in range(k)]
    for p in paths:
        if os.path.exists(p):
            os.remove(p)
    step = n // k

    def one(i):
        flags = os.O_WRONLY | os.O_CREAT | (os.O_DIRECT if direct else 0)
        fd = os.open(paths[i], flags, 0o644)
        try:
            for lo in range(i * step, (i + 1) * step, 16 << 20):
                os.write(fd, memoryview(buf[lo:lo + (16 << 20)]))
        finally:
            os.close(fd)
    t = time.perf_counter()
    with ThreadPoolExecutor(k) as ex:
        list(ex.map(one, range(k)))
    dt = time.perf_counter() - t
    for p in paths:
        os.remove(p)
    return dt


for root in (tmp, '/dev/shm'):
    for k in (1, 2, 4, 8):
        try:
            ts = [many_files(root, k) for _ in range(3)]
            print(json.dumps({"case": "{} files at once, one thread each, in {}".format(k, root),
                              "GBps": round(n / min(ts) / 1e9, 2), "all_s": [round(t, 3) for t in ts]}), flush=True)
        except Exception as exc:
            print(json.dumps({"case": "{} files in {}".format(k, root), "error": repr(exc)[:200]}), flush=True)
for k in (1, 4):
    try:
        ts = [many_files(tmp, k, direct=True) for _ in range(2)]
        print(json.dumps({"case": "O_DIRECT, {} files at once in {}".format(k, tmp),
                          "GBps": round(n / min(ts) / 1e9, 2), "all_s": [round(t, 3) for t in ts]}), flush=True)
    except Exception as exc:
        print(json.dumps({"case": "O_DIRECT {} files".format(k), "error": repr(exc)[:200]}), flush=True)
