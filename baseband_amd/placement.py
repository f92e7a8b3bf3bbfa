"""Where the decoded output lies in HBM.

The write rate of a decode launch depends on how its output was allocated more
than on anything inside the kernels (docs/DESIGN_rounds1-3.md section 3.2): into a plain
allocation of 4-34 GB -- the output of an ordinary ``read()`` -- the same
launch runs at 5.3-5.7 TB/s in most draws and at 6.4-6.8 in some, while every
tensor cut from an arena of VMM chunks (`baseband_amd.arena`) decodes at
6.5-6.8 (profiles/r03f_exp_arena.log, r03g_exp_arena_*.log, r03h_*).

`empty_output` is what the readers allocate their outputs with: from the
process-wide arena for outputs of `ARENA_MIN_BYTES` to `ARENA_MAX_BYTES`, else
(and whenever the arena cannot serve) ``torch.empty``.  File bytes kept in HBM
(``dtype=torch.uint8``: the staged copy of a file, `fh.stage()`) come from it
too: the same launches run 1-2 % faster when their INPUT lies in arena memory
(profiles/r03u_exp_image_arena.log, r03u_exp_headline_alloc.log).  The arena is created on
first use; it is only a virtual range until tensors need memory, and
`release_unused()` gives unused memory back (done automatically when torch
runs out of memory here, and once the arenas have held no block for
BB_ARENA_IDLE_S = 30 seconds with no reader open).  One arena per device.

    BB_ARENA=0          never create one (plain ``torch.empty`` everywhere)
    BB_ARENA_GIB=<n>    capacity of the arena in GiB (default: the device's memory)
    BB_ARENA_IDLE_S=<s> seconds without a live block or an open reader after which the
                        arenas give their memory back (default 30; 0 = at once)
    BB_ARENA_KEEP=1     never give memory back automatically
    BB_ARENA_PREPARE=0  do not start growing when a reader of a large stream is opened
    baseband_amd.arena.enable(capacity) / .disable()   the same from the program

Round 2's `empty_output(shape, candidates=k)` -- allocate k tensors, probe
each, keep the fastest -- is gone: it needed k times the memory and still lost
when all k draws were slow.
"""
import os
import sys
import threading
import time
import warnings

import numpy as np
import torch

from . import arena as _arena

__all__ = ['empty_output', 'prepare_output', 'release_unused', 'ARENA_MIN_BYTES', 'ARENA_MAX_BYTES']

# smaller outputs come from torch's allocator: where they lie changes a launch
# of 0.15 ms by tens of microseconds at most, and the arena takes the device's
# memory in steps of 48 GiB -- a script that only reads small pieces (and its
# 64 MiB read-ahead window) should not make it take one
ARENA_MIN_BYTES = 1 << 30
# larger outputs as well: a plain allocation of that size spans most of the
# device anyway and decodes as fast or faster (127.5 GiB, one combination per
# process, three processes each: plain 0.82-0.85 of the peak, arena 0.73-0.82;
# profiles/r03u_exp_headline_alloc.log)
ARENA_MAX_BYTES = 64 << 30
_failed = False                 # arena creation failed once: do not try again in this process
_ITEMSIZE = {torch.float32: 4, torch.complex64: 8, torch.uint8: 1, torch.int32: 4, torch.int64: 8, torch.int8: 1,
             torch.float64: 8, torch.float16: 2, torch.bfloat16: 2, torch.int16: 2, torch.bool: 1, torch.complex128: 16}


_open_readers = 0               # stream readers open right now (GPUStreamReaderBase registers itself)
_count_lock = threading.Lock()


def reader_opened():
    global _open_readers
    with _count_lock:
        _open_readers += 1


def reader_closed():
    """The last reader to close gives unused arena memory back."""
    global _open_readers
    with _count_lock:
        _open_readers = max(0, _open_readers - 1)
        last = _open_readers == 0
    if last:
        _auto_trim()


_idle_deadline = None            # time.monotonic() after which the watcher trims (None: nothing scheduled)
_idle_watcher = None             # the daemon thread waiting for that deadline
_idle_lock = threading.Lock()


_idle_trims = 0                  # idle trims that released memory so far (each doubles the next delay)


def _idle_seconds():
    """Seconds of "nothing alive" after which the arenas trim: BB_ARENA_IDLE_S
    (default 30), DOUBLED by every idle trim that released memory, up to an
    hour.  Every trim + regrow burns a step's worth of virtual addresses, which
    are never reused (csrc/bb_arena.inc: about 2,700 steps of 48 GiB per
    process): a service whose bursts lie a minute apart would use them up in
    two days at a fixed 30 s; with the doubling it soon keeps its memory between
    bursts and trims once an hour of idleness at most (112 days of that)."""
    try:
        base = float(os.environ.get('BB_ARENA_IDLE_S', '30'))
    except ValueError:
        base = 30.0
    return min(base * 2 ** min(_idle_trims, 12), max(base, 3600.0))


def _nothing_alive():
    return not _open_readers and all(a.live_blocks() == 0 for a in _arena.all_arenas() if a._handle)


def _trim_idle():
    """Trim the arenas if STILL nothing needs their memory."""
    if os.environ.get('BB_ARENA_KEEP', '0') not in ('0', '', 'no', 'off') or not _nothing_alive():
        return 0
    if sys.is_finalizing():         # (the watcher thread must not be inside hipMemUnmap while the interpreter goes)
        return 0
    global _idle_trims
    freed = 0
    for a in _arena.all_arenas():
        try:
            if a._handle and a.live_blocks() == 0 and a.stats()['bytes_backed']:
                with torch.cuda.device(a.device):       # (this may be the watcher thread: its current device is 0)
                    freed += a.trim()
        except Exception:
            pass
    if freed:
        _idle_trims += 1
    return freed


def _watch_idle():
    """Body of the watcher thread: sleep until the deadline (which later frees
    push back), trim, leave."""
    global _idle_deadline, _idle_watcher
    while True:
        with _idle_lock:
            wait = None if _idle_deadline is None else _idle_deadline - time.monotonic()
            if wait is None or wait <= 0:
                _idle_deadline = None
                _idle_watcher = None
                due = wait is not None
                break
        time.sleep(min(wait, 1.0))
    if due:
        try:
            _trim_idle()
        except Exception:
            pass


def _auto_trim(ar=None):
    """A block died or the last reader closed: when no reader is open and no
    arena holds a live block, give the memory back -- after BB_ARENA_IDLE_S
    seconds (default 30) in which that stays so.  Not at
    once, and not after a few seconds: growing again is not cheap -- memory
    that was released before is cleared by the driver when it is created
    again, 1.5 s for a 48 GiB step on some boxes and 4.6 s on others
    (profiles/r04h_prof_arena_grow.log, r04zzz_bench.json `mid_size.arena.grow_ms`),
    and decodes 6 % slower afterwards (0.77 against 0.81-0.84 of the peak into
    memory that was never used) -- and a script that reads file after file
    drops to "nothing alive" between two reads all the time.  Costs a
    clock reading per call: ONE daemon thread waits for the deadline, which
    every later call pushes back.  BB_ARENA_IDLE_S=0: at once; BB_ARENA_KEEP=1: never."""
    global _idle_deadline, _idle_watcher
    if _open_readers or os.environ.get('BB_ARENA_KEEP', '0') not in ('0', '', 'no', 'off'):
        return 0
    delay = _idle_seconds()
    if delay <= 0:
        return _trim_idle() if _nothing_alive() else 0
    with _idle_lock:
        _idle_deadline = time.monotonic() + delay
        if _idle_watcher is None:
            _idle_watcher = threading.Thread(target=_watch_idle, name='bb-arena-idle', daemon=True)
            _idle_watcher.start()
    return 0


def _arena_for(device, create=True):
    """The readers' arena on `device` (one per device; created now if there is
    none yet and `create` allows); None when switched off or not available."""
    global _failed
    if not isinstance(device, torch.device):
        device = torch.device(device)
    ar = _arena._arenas.get(device.index) if device.index is not None else _arena.default(device)
    if ar is not None:
        return ar
    if not create or _failed or os.environ.get('BB_ARENA', '1') in ('0', 'off', 'no'):
        return None
    env = os.environ.get('BB_ARENA_GIB')
    try:
        ar = _arena.get_or_create(device, int(float(env) * 2 ** 30) if env else None)
        ar.on_block_freed = _auto_trim
        return ar
    except Exception as exc:            # no VMM on this system: the readers work without it
        _failed = True
        warnings.warn("baseband_amd: no output arena ({!r}); outputs come from torch.empty".format(exc))
        return None


def prepare_output(nbytes, device=None):
    """A reader has been opened whose whole decoded stream is `nbytes`: if that
    is an output the arena would hold (`ARENA_MIN_BYTES` and more; capped at
    `ARENA_MAX_BYTES`) and the arena has no room for it, start growing in the
    background (`Arena.prepare`), so that the step's creation runs next to the
    rest of ``open()``, the header scan and the first staging windows instead of
    inside the first ``read()``.  BB_ARENA_PREPARE=0 switches
    it off.  Never raises; returns True if the arena was asked."""
    try:
        if nbytes < ARENA_MIN_BYTES or os.environ.get('BB_ARENA_PREPARE', '1') in ('0', 'off', 'no'):
            return False
        if not torch.cuda.is_available():
            return False
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        ar = _arena_for(device, True)
        return bool(ar is not None and ar.prepare(min(int(nbytes), ARENA_MAX_BYTES)))
    except Exception:
        return False


_slow_allocation_warned = False


def _note_slow_allocation(t0, nbytes):
    """Say ONCE per process where seconds of a first large read went when they went into
    taking memory: device memory that was freed a moment ago is wiped by the driver when
    it is handed out again -- 22 ms per GiB freed, 6 s after a 275 GiB release
    (profiles/r05f_cold_read.json) -- and whoever allocates next pays, here
    the reader's output.  Nothing user space can shorten; the warning makes it attributable
    (VERDICT r5 next 8)."""
    global _slow_allocation_warned
    if t0 is None or _slow_allocation_warned:
        return
    waited = time.perf_counter() - t0
    if waited > 1.0:
        _slow_allocation_warned = True
        warnings.warn("baseband_amd: allocating {:.1f} GiB of output took {:.1f} s -- the GPU driver was clearing "
                      "device memory that had been freed shortly before (about 22 ms per GiB freed); the decode itself "
                      "is not slower.  Keep large tensors alive, or reuse them with read(out=...), to avoid it."
                      .format(nbytes / 2 ** 30, waited), RuntimeWarning, stacklevel=3)


def release_unused(device=None):
    """Give the unused physical memory of the arenas (of `device`, default: of
    every device) back to the device; bytes.  Call it before a large
    allocation of your own next to open readers; the readers do it themselves
    when THEIR allocation runs out of memory, and arenas trim by themselves
    once nothing of theirs is alive and no reader is open."""
    if device is not None:
        ar = _arena.default(device)
        return ar.trim() if ar is not None else 0
    return sum(a.trim() for a in _arena.all_arenas())


def empty_output(shape, dtype=torch.float32, device=None, create=True):
    """Uninitialised device tensor like ``torch.empty(shape, dtype=dtype,
    device='cuda')`` for a decode launch to write into: from the arena when the
    output is large, else from torch's allocator.  ``create=False``: only from
    an arena that exists already (the readers' copies of file bytes: a staged
    file alone should not make the arena take its first 48 GiB step)."""
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    elif not isinstance(device, torch.device) or device.index is None:
        device = torch.device(device)
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
    shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
    item = _ITEMSIZE.get(dtype)
    if item is None:
        item = torch.empty(0, dtype=dtype).element_size()
    nbytes = item
    for s_ in shape:
        nbytes *= s_
    t0 = time.perf_counter() if nbytes >= ARENA_MIN_BYTES else None
    if ARENA_MIN_BYTES <= nbytes <= ARENA_MAX_BYTES:
        ar = _arena_for(device, create)
        if ar is not None:
            t = ar.empty(shape, dtype)
            if t is not None:
                _note_slow_allocation(t0, nbytes)
                return t
    try:
        t = torch.empty(shape, dtype=dtype, device=device)
        _note_slow_allocation(t0, nbytes)
        return t
    except torch.cuda.OutOfMemoryError:
        if not release_unused():
            raise
    # memory the arena gave back a moment ago is not allocatable until the
    # driver has cleared it: a little patience before the error counts
    for attempt in range(8):
        try:
            return torch.empty(shape, dtype=dtype, device=device)
        except torch.cuda.OutOfMemoryError:
            if attempt == 7:
                raise
            torch.cuda.empty_cache()
            time.sleep(0.25)
