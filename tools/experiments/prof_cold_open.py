"""Where do the ~140 ms between a cold and a warm first large read go
(bench_legs/cold_read.py)?  Times, in a fresh process:
HIP context, library init, arena creation, a background prepare, the first
block, the pinned staging buffers, the first mapping of a file."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C      # noqa: E402
import numpy as np      # noqa: E402
import torch            # noqa: E402


def ms(t0):
    return round((time.perf_counter() - t0) * 1e3, 2)


t0 = time.perf_counter(); torch.zeros(1, device='cuda'); torch.cuda.synchronize(); print("hip context", ms(t0))
from baseband_amd import kernels, arena, placement, _lib, staging   # noqa: E402
t0 = time.perf_counter(); kernels.init(); print("kernels.init", ms(t0))
t0 = time.perf_counter(); props = torch.cuda.get_device_properties(0); print("get_device_properties", ms(t0))
t0 = time.perf_counter(); free_b, total_b = torch.cuda.mem_get_info(); print("mem_get_info", ms(t0))
h = C.c_void_p()
t0 = time.perf_counter(); rc = _lib.lib.bb_arena_create(props.total_memory, C.byref(h)); print("bb_arena_create (raw)", ms(t0), rc)
s = _lib.ArenaStats(); _lib.lib.bb_arena_get_stats(h, C.byref(s)); print("   create_ms inside", round(s.create_ms, 3), "va_reserved TiB", s.va_reserved / 2 ** 40)
t0 = time.perf_counter(); rc = _lib.lib.bb_arena_prepare(h, 34 << 30); print("bb_arena_prepare call", ms(t0), rc)
p = C.c_void_p()
t0 = time.perf_counter(); rc = _lib.lib.bb_arena_alloc(h, 34 << 30, C.byref(p)); print("bb_arena_alloc (waits for the step)", ms(t0), rc)
_lib.lib.bb_arena_get_stats(h, C.byref(s)); print("   prepare_ms", round(s.prepare_ms, 2), "wait", round(s.prepare_wait_ms, 2), "grow_ms", round(s.grow_ms, 2))
_lib.lib.bb_arena_free(h, p)
t0 = time.perf_counter(); _lib.lib.bb_arena_trim(h, None); print("trim", ms(t0))
t0 = time.perf_counter(); rc = _lib.lib.bb_arena_alloc(h, 34 << 30, C.byref(p)); print("bb_arena_alloc (grows itself, probed)", ms(t0), rc)
_lib.lib.bb_arena_free(h, p)
t0 = time.perf_counter(); _lib.lib.bb_arena_destroy(h); print("destroy", ms(t0))
# a second arena: is the first creation special?
t0 = time.perf_counter(); rc = _lib.lib.bb_arena_create(props.total_memory, C.byref(h)); print("bb_arena_create again", ms(t0), rc)
_lib.lib.bb_arena_destroy(h)
t0 = time.perf_counter(); ar = arena.Arena(props.total_memory); print("arena.Arena()", ms(t0))
ar.close()
for k in range(3):
    t0 = time.perf_counter(); b = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True); print("pinned 64 MiB", ms(t0))
t0 = time.perf_counter(); st = torch.cuda.Stream(); print("side stream", ms(t0))
t0 = time.perf_counter(); e = torch.cuda.Event(); e.record(); print("event", ms(t0))
