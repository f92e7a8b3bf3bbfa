import sys, time, torch
sys.path.insert(0, '/root/repo')
from baseband_amd import arena, kernels
kernels.init()
ar = arena.Arena(250 << 30)
for k in range(3):
    t0 = time.perf_counter(); t = ar.empty(1 << 28); torch.cuda.synchronize(); t1 = time.perf_counter()
    del t
    r = ar.trim(); t2 = time.perf_counter()
    print("grow %.3f s, trim %.3f s (%d GiB)" % (t1 - t0, t2 - t1, r >> 30), flush=True)
    time.sleep(1.0)
