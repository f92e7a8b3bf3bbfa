import sys, time, torch
sys.path.insert(0, '/root/repo')
from baseband_amd import arena, kernels, placement
import baseband_amd
kernels.init()
ar = arena.Arena(100 << 30)
t = ar.empty(1 << 28); del t
def bench(fn, n=2000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("arena.empty(1 GiB) + del: %.1f us" % bench(lambda: ar.empty(1 << 28)))
print("torch.empty(1 GiB) + del: %.1f us" % bench(lambda: torch.empty(1 << 28, device='cuda')))
import ctypes as C
from baseband_amd._lib import lib
p = C.c_void_p()
def raw():
    lib.bb_arena_alloc(ar._handle, 1 << 30, C.byref(p)); lib.bb_arena_free(ar._handle, p)
print("bb_arena_alloc + free via ctypes: %.1f us" % bench(raw))
blk = arena._Block(None, p.value or 0, 1 << 30, (1 << 28,), '<f4', torch.cuda.current_stream())
lib.bb_arena_alloc(ar._handle, 1 << 30, C.byref(p)); blk.ptr = p.value
blk.__cuda_array_interface__['data'] = (p.value, False)
print("torch.as_tensor(block): %.1f us" % bench(lambda: torch.as_tensor(blk, device='cuda')))
ev = torch.cuda.Event()
print("Event() + record: %.1f us" % bench(lambda: torch.cuda.Event().record()))
print("current_stream: %.1f us" % bench(lambda: torch.cuda.current_stream()))
print("empty_output 1 GiB: %.1f us" % bench(lambda: baseband_amd.empty_output((1 << 28,))))
