"""Round 4: what the page cache -> pinned copy and the teardown of a file mapping cost on the GPU box:
copy rate from a FRESH mapping (pages fault in) and from a populated one, by copy threads; munmap time.
    python tools/prof_munmap.py"""
import mmap, os, sys, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
size = 2 << 30
path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'bb_mm_test.bin')
blk = np.random.default_rng(0).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
with open(path, 'wb') as f:
    for _ in range(size // len(blk)):
        f.write(blk)
dst = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True).numpy()
W = 64 << 20


def copy_all(arr, pool, nthr):
    t0 = time.perf_counter()
    step = W // nthr
    for lo in range(0, size, W):
        futs = [pool.submit(np.copyto, dst[o:o + step], arr[lo + o:lo + o + step]) for o in range(0, W, step)]
        for f_ in futs:
            f_.result()
    return time.perf_counter() - t0


for nthr in (4, 8, 16, 32, 64):
    pool = ThreadPoolExecutor(nthr)
    f = open(path, 'rb')
    mm = mmap.mmap(f.fileno(), size, access=mmap.ACCESS_READ)
    arr = np.frombuffer(mm, np.uint8)
    t_fresh = copy_all(arr, pool, nthr)
    t_again = copy_all(arr, pool, nthr)
    t0 = time.perf_counter(); del arr; mm.close(); t_un = time.perf_counter() - t0
    f.close()
    # MAP_POPULATE
    f = open(path, 'rb')
    t0 = time.perf_counter()
    mm = mmap.mmap(f.fileno(), size, flags=mmap.MAP_SHARED | mmap.MAP_POPULATE, prot=mmap.PROT_READ)
    t_pop = time.perf_counter() - t0
    arr = np.frombuffer(mm, np.uint8)
    t_popcopy = copy_all(arr, pool, nthr)
    del arr; mm.close(); f.close()
    pool.shutdown()
    print("threads %2d: fresh mapping %.1f GB/s, populated %.1f GB/s, munmap %.1f ms; MAP_POPULATE %.1f ms then %.1f GB/s"
          % (nthr, size / t_fresh / 1e9, size / t_again / 1e9, t_un * 1e3, t_pop * 1e3, size / t_popcopy / 1e9), flush=True)

os.remove(path)
