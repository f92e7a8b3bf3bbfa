"""Output memory for the decode launches (include/bbdecode_arena.h).

The write rate of a decode launch depends on how its output was allocated
(docs/DESIGN_rounds1-3.md 3.2): into plain allocations of 4-34 GB it is 5.3-5.7 TB/s in most
draws and 6.4-6.8 in some; into memory created as chunks with the HIP virtual
memory API and mapped into one virtual range it is 6.5-6.8 in every draw.  An
`Arena` is such a range: backed on demand by 32 MiB chunks, first-fit
allocator on top.  Tensors come from `Arena.empty()`; they alias arena memory
through ``__cuda_array_interface__`` and give their block back when the last
view of them dies.  `trim()` returns unused physical memory to the device;
the per-device arenas of the readers (`default(device)`) trim themselves once
they have held no block for a few seconds with no reader open (`placement`): a
program that has finished reading does not sit on a 48 GiB step.  Unmapping waits first for the work that was
queued on freed blocks (the events recorded at each free).

Streams: a block goes back to the arena when the last tensor viewing it is
garbage collected -- at once, without waiting for the work queued on it (the
readers return before their decode has finished).  Reuse on the SAME stream is
ordered by the stream itself; when memory of a freed block is handed out on
ANOTHER stream, that stream is first made to wait (``wait_event``, no host
sync) for everything that had been queued, at the time of the free, on the
stream the block was allocated on.  As with torch's own allocator, a tensor
that is USED on a stream other than the one it was allocated on is the
caller's to order.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def _current_stream(device):
    from .kernels import current_stream         # (kernels imports nothing of this module)
    return current_stream(device)


class _Block:
    """Owner of one arena block; torch keeps it alive through the array
    interface and drops it with the last tensor that views the block."""
    __slots__ = ('arena', 'ptr', 'nbytes', 'stream', '__cuda_array_interface__', '__weakref__')

    def __init__(self, arena, ptr, nbytes, shape, typestr, stream):
        self.arena, self.ptr, self.nbytes, self.stream = arena, ptr, nbytes, stream
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (ptr, False),
                                         'version': 2, 'strides': None}

    def __del__(self):
        arena = self.arena
        if arena is None or not arena._handle:
            return
        # under the arena's lock: a trim on another thread (placement's idle
        # watcher) either sees this free -- and waits for the event recorded
        # here before it unmaps -- or has finished before the block counts as
        # free (between its wait and its unmap nothing may slip in)
        try:
            lock = arena._lock
            lock.acquire()
        except Exception:           # interpreter shutdown
            return
        try:
            try:
                arena._note_free(self)
            except Exception:       # interpreter shutdown: torch may be half gone
                pass
            try:
                lib.bb_arena_free(arena._handle, C.c_void_p(self.ptr))
            except Exception:       # ... and so may this module's globals
                return
        finally:
            lock.release()
        hook = arena.on_block_freed
        if hook is not None:
            try:
                hook(arena)
            except Exception:
                pass


_TYPESTR = {torch.float32: '<f4', torch.uint8: '|u1', torch.int32: '<i4', torch.int64: '<i8',
            torch.int8: '|i1', torch.float64: '<f8'}
_ITEMSIZE = {torch.float32: 4, torch.uint8: 1, torch.int32: 4, torch.int64: 8, torch.int8: 1, torch.float64: 8}


class Arena:
    """``Arena(capacity_bytes)`` on the current (or given) device.  Nothing is
    taken from the device until tensors are asked for."""

    def __init__(self, capacity, device=None):
        from . import kernels
        kernels.require_gpu()
        self._handle = None
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            kernels.init()
            check(lib.bb_arena_create(int(capacity), C.byref(h)), 'bb_arena_create')
        self._handle = h
        st = self.stats()
        self._granule = int(st['chunk_bytes'])
        self._freed = []            # [(lo, hi, event, stream)] of blocks freed with work possibly in flight
        self._lock = threading.RLock()      # (re-entrant: a block's __del__ may run while it is held)
        self.on_block_freed = None  # called with the arena after a block went back (placement: auto trim)

    def _note_free(self, block):
        """Remember what was queued on the block's stream when it was freed."""
        ev = torch.cuda.Event()
        ev.record(block.stream)
        hi = block.ptr + -(-block.nbytes // self._granule) * self._granule
        with self._lock:
            if len(self._freed) >= 64:
                self._freed = [f for f in self._freed if not f[2].query()]
            self._freed.append((block.ptr, hi, ev, block.stream))

    def _order_reuse(self, lo, hi, stream):
        """Memory [lo, hi) is about to be used on `stream`: wait there for the
        work that was queued on OTHER streams when blocks inside it were freed."""
        with self._lock:
            hits = [f for f in self._freed if f[0] < hi and lo < f[1]]
            if hits:        # (an entry only partly covered stays: the rest of it may go to another stream)
                self._freed = [f for f in self._freed if not (lo <= f[0] and f[1] <= hi)]
        for flo, fhi, ev, fstream in hits:
            if fstream != stream:
                stream.wait_event(ev)

    def empty(self, shape, dtype=torch.float32):
        """Uninitialised tensor of `shape` in arena memory, or None when
        neither the arena's capacity nor the device's free memory allow it (the
        caller falls back to ``torch.empty``).  complex64 is allocated as
        float32 pairs."""
        if self._handle is None:
            return None
        # (this runs once per read(): plain Python arithmetic, no NumPy / torch helpers --
        # 26 -> about 12 us, tools/experiments/prof_arena_empty.py)
        shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
        cplx = dtype == torch.complex64
        base = torch.float32 if cplx else dtype
        n = 2 if cplx else 1
        for s_ in shape:
            n *= s_
        if n == 0:
            return torch.empty(shape, dtype=dtype, device=self.device)
        item = _ITEMSIZE.get(base)
        if item is None:
            return None                 # (checked BEFORE a block is taken: the caller uses torch.empty)
        p = C.c_void_p()
        if torch.cuda.current_device() == self.device.index:
            rc = lib.bb_arena_alloc(self._handle, n * item, C.byref(p))
        else:
            with torch.cuda.device(self.device):
                rc = lib.bb_arena_alloc(self._handle, n * item, C.byref(p))
        if rc == _lib.BB_ERANGE or not p.value:
            return None
        check(rc, 'bb_arena_alloc')
        stream = _current_stream(self.device)
        nbytes = n * item
        self._order_reuse(p.value, p.value + -(-nbytes // self._granule) * self._granule, stream)
        block = _Block(self, p.value, nbytes, (n,), _TYPESTR[base], stream)
        t = torch.as_tensor(block, device=self.device)
        if cplx:
            t = torch.view_as_complex(t.view(-1, 2))
        return t.view(shape)

    def owns(self, tensor):
        """Does `tensor` live in this arena?"""
        return bool(self._handle is not None and tensor.is_cuda
                    and lib.bb_arena_owns(self._handle, C.c_void_p(tensor.data_ptr())))

    def prepare(self, nbytes):
        """Start growing NOW, on a thread of the library, for a block of
        `nbytes` that will be asked for soon (`bb_arena_prepare`): returns at
        once; the next `empty()` that needs the step waits only for what is
        left of its creation.  True if a growth was started or none is needed."""
        if self._handle is None:
            return False
        if torch.cuda.current_device() == self.device.index:
            rc = lib.bb_arena_prepare(self._handle, int(nbytes))
        else:
            with torch.cuda.device(self.device):
                rc = lib.bb_arena_prepare(self._handle, int(nbytes))
        return rc == _lib.BB_OK

    def _wait_freed(self):
        """Host-wait for everything that was queued on blocks at the time they
        were freed: their memory is about to be unmapped (bb_arena_trim /
        bb_arena_destroy), and a kernel still writing there would fault
        (ADVICE r3).  Live blocks are never unmapped by a trim."""
        with self._lock:
            pending, self._freed = self._freed, []
        for _, _, ev, _ in pending:
            try:
                ev.synchronize()
            except Exception:           # interpreter shutdown
                pass

    def live_blocks(self):
        return int(self.stats()['blocks']) if self._handle else 0

    def trim(self):
        """Give physical memory that no live tensor uses back to the device
        (whole growth steps from the end of the backed part); returns bytes.
        Waits for the work queued on freed blocks first."""
        if self._handle is None:
            return 0
        n = C.c_size_t()
        with self._lock:                # (frees wait: see _Block.__del__)
            if self._handle is None:
                return 0
            self._wait_freed()
            check(lib.bb_arena_trim(self._handle, C.byref(n)), 'bb_arena_trim')
        return int(n.value)

    def stats(self):
        s = _lib.ArenaStats()
        check(lib.bb_arena_get_stats(self._handle, C.byref(s)), 'bb_arena_get_stats')
        out = {f: getattr(s, f) for f, _ in s._fields_}
        out['probe_history'] = [int(v) for v in s.probe_history[:s.probe_history_n]]    # GB/s, oldest first
        del out['probe_history_n']
        return out

    def close(self):
        """Release the arena's memory.  Tensors still alive become invalid."""
        with self._lock:
            h, self._handle = self._handle, None
            if h:
                self._wait_freed()
                try:
                    torch.cuda.synchronize(self.device)     # (work on still-live tensors, which become invalid)
                except Exception:
                    pass
                check(lib.bb_arena_destroy(h), 'bb_arena_destroy')

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_arenas = {}                    # device index -> the readers' arena on that device
_arenas_lock = threading.Lock()


def _index(device):
    if device is None:
        return torch.cuda.current_device()
    device = torch.device(device)
    return torch.cuda.current_device() if device.index is None else device.index


def enable(capacity=None, device=None):
    """Create the arena the readers take their output tensors from on
    `device` (default: the current one) and return it, REPLACING one that
    exists there (its tensors become invalid: only the program itself should
    call this); `capacity` None: the whole memory of the device (virtual range
    only: physical memory follows the tensors).  ``disable()`` drops it."""
    idx = _index(device)
    with _arenas_lock:
        old = _arenas.pop(idx, None)
    if old is not None:
        old.close()
    if capacity is None:
        capacity = torch.cuda.mem_get_info(idx)[1]
    ar = Arena(capacity, device=idx)
    with _arenas_lock:
        _arenas[idx] = ar
    return ar


def get_or_create(device, capacity=None):
    """The arena of `device`, created if there is none yet -- under a lock and
    WITHOUT replacing one another thread made meanwhile (lazy creation by the
    readers: two first reads on two threads share one arena)."""
    idx = _index(device)
    with _arenas_lock:
        ar = _arenas.get(idx)
        if ar is None:
            if capacity is None:
                # (not get_device_properties: its first call takes 107 ms, tools/experiments/prof_cold_open.py)
                capacity = torch.cuda.mem_get_info(idx)[1]
            ar = _arenas[idx] = Arena(capacity, device=idx)
        return ar


def disable(device=None):
    """Drop the arena of `device`; None: the arenas of every device."""
    with _arenas_lock:
        if device is None:
            gone = list(_arenas.values())
            _arenas.clear()
        else:
            a = _arenas.pop(_index(device), None)
            gone = [a] if a is not None else []
    for a in gone:
        a.close()


def default(device=None):
    """The readers' arena on `device` (default: the current device), or None."""
    if not _arenas:
        return None
    return _arenas.get(_index(device))


def all_arenas():
    with _arenas_lock:
        return list(_arenas.values())
