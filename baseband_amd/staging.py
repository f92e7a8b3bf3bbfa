"""Host file image -> HBM staging with a pinned, double-buffered pipeline.

The reference reads a frame at a time with ``fh.read`` / ``np.memmap``
(base/payload.py:122-137).  Here a file is moved in large windows (an integer
number of frame sets, tens of MiB): the CPU copies window k+1 from the page
cache into a pinned buffer while the DMA engine moves window k on a side
stream (hipMemcpyAsync) and the compute stream scans/decodes window k-1.
Ordering is by events only; nothing blocks the device.
"""
import atexit
import ctypes
import mmap
import os
import queue
import threading
import time
import weakref
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from .placement import empty_output


def host_image(fh):
    """uint8 view of a whole file: memory-mapped for real files, a byte copy
    for in-memory streams.  The file position is left unchanged."""
    if hasattr(fh, 'host_image'):            # helpers.sequentialfile reader
        return fh.host_image()
    try:
        fileno = fh.fileno()
    except Exception:
        fileno = None
    if fileno is not None:
        size = os.fstat(fileno).st_size
        if size == 0:
            return np.empty(0, dtype=np.uint8)
        mm = mmap.mmap(fileno, size, access=mmap.ACCESS_READ)
        img = np.frombuffer(mm, dtype=np.uint8).view(FileImage)
        img.fd = fileno                     # (BB_STAGE_WINDOW_MMAP: large windows get a mapping of their own)
        st = os.fstat(fileno)
        img.fd_id = (st.st_dev, st.st_ino, st.st_size)      # ... if the number still names THIS file then
        img.mm = mm                         # (`retire_image` empties the mapping's page tables in the background)
        return img
    pos = fh.tell()
    fh.seek(0)
    data = fh.read()
    fh.seek(pos)
    return np.frombuffer(data, dtype=np.uint8)


class FileImage(np.ndarray):
    """The mapped bytes of a real file, remembering its descriptor.  With
    BB_STAGE_WINDOW_MMAP=1 the windows of a large read are copied out of
    SHORT-LIVED mappings of their own (`_stage`) instead of this whole-file
    mapping, whose pages then never get populated: tearing down a mapping costs
    per populated page -- 6 ms for a 2 GiB file that was read through it, paid
    at close() (since late in round 4 on a background thread: `retire_image`) --
    while a window's mapping is torn down inside the pipeline's loop.  Measured (profiles/r04w_pipeline_window_mmap.log):
    open() 6-7 -> 0.4 ms, but the page-cache copies slow down (fresh mappings
    fault their pages in while the previous window's is being torn down) and
    the totals scatter over each other on this host: OFF by default."""
    fd = None
    fd_id = None
    mm = None

    def __array_finalize__(self, obj):
        self.fd = None                      # (views and slices are plain arrays as far as staging cares)
        self.fd_id = None
        self.mm = None


_reaper = None
_RETIRE = os.environ.get('BB_STAGE_RETIRE', '1') not in ('0', 'no', 'off')
_RETIRE_MIN = int(os.environ.get('BB_STAGE_RETIRE_MIN_MIB', 8)) << 20      # smaller mappings: ordinary teardown (3 us per MiB)


def _zap(keep, addr, n):
    # madvise through ctypes, which drops the GIL for the call (mmap.madvise
    # keeps it: the reader's thread then waits a GIL switch interval, 5 ms, at
    # its next bytecode -- profiles/r04zy_prof_close.log), in pieces of 16 MiB:
    # MADV_DONTNEED holds the address space lock for READING, which page faults
    # of a read that runs meanwhile share, but an mmap() inside the HIP runtime
    # (the next reader's first copy on a new stream) waits for -- 6.8 ms behind
    # one call over 2 GiB, 50 us behind a piece.  `keep` (the mmap object and a
    # view of it, which also bars an explicit close()) is only held so that the
    # mapping outlives the calls.
    global _madvise
    try:
        if _madvise is None:
            libc = ctypes.CDLL(None, use_errno=True)
            libc.madvise.argtypes = (ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)
            libc.madvise.restype = ctypes.c_int
            _madvise = libc.madvise
        step = _ZAP_STEP
        for lo in range(0, n, step):
            _madvise(addr + lo, min(step, n - lo), mmap.MADV_DONTNEED)
    except Exception:
        pass


_ZAP_STEP = int(os.environ.get('BB_STAGE_RETIRE_STEP_MIB', 16)) << 20
_madvise = None


def retire_mapping(mm, view):
    """Empty the page tables of mapping `mm` (of which `view` is a uint8 array
    from offset 0) on the background thread; mappings below 8 MiB are left to
    the ordinary teardown."""
    global _reaper
    if not _RETIRE or mm is None or len(view) < _RETIRE_MIN:
        return
    if _reaper is None:
        _reaper = ThreadPoolExecutor(1, thread_name_prefix='bb-retire')
    _reaper.submit(_zap, (mm, view), view.ctypes.data, len(view))


def retire_image(img):
    """A reader is done with the whole-file mapping `img` (from `host_image`;
    a `SequenceImage` retires the mappings of its files): empty its page tables
    on a background thread.  Tearing down a mapping costs per populated page --
    6 ms for a 2 GiB file that was read through it
    (profiles/r04x_prof_munmap.log) -- and without this close() pays for it when
    the last view of the mapping goes (a loop over 2 GiB files: 0.80 -> 0.93 of
    the link, profiles/r04zx_pipeline_retire.log).
    The pages stay in the page cache; views of the image that are still around
    simply fault them in again."""
    if hasattr(img, 'retire'):
        img.retire()
        return
    retire_mapping(getattr(img, 'mm', None), img)


_WINDOW_MMAP = os.environ.get('BB_STAGE_WINDOW_MMAP', '0') not in ('0', 'no', 'off')
_GRAN = mmap.ALLOCATIONGRANULARITY
_COPY_THREADS = int(os.environ.get('BB_COPY_THREADS', 0)) or max(1, min(8, (os.cpu_count() or 2) // 2))
_copy_pool = None


def _parallel_copy(dst, src):
    """dst[:] = src with several threads (NumPy releases the GIL while
    copying): one core moves ~10 GB/s from the page cache, the host link
    wants ~50 GB/s."""
    global _copy_pool
    n = len(src)
    if n < (8 << 20) or _COPY_THREADS == 1:
        dst[:n] = src
        return
    if _copy_pool is None:
        _copy_pool = ThreadPoolExecutor(_COPY_THREADS)
    step = -(-n // _COPY_THREADS)
    step += -step % 4096
    futs = [_copy_pool.submit(np.copyto, dst[o:min(n, o + step)], src[o:min(n, o + step)])
            for o in range(0, n, step)]
    for f in futs:
        f.result()


def _stage(dst, image, lo, hi):
    """dst[:hi-lo] = image[lo:hi]; images made of several mappings (a file
    sequence, the raw files of a GSB observation) are copied piece by piece,
    straight from each mapping, the pieces spread over the copy threads."""
    global _copy_pool
    if not hasattr(image, 'pieces'):
        fd = getattr(image, 'fd', None) if _WINDOW_MMAP else None
        if fd is not None and hi - lo >= (4 << 20):
            try:
                # the number may have been closed and handed to another file since the
                # image was made: map a window only from the file recorded
                st = os.fstat(fd)
                if (st.st_dev, st.st_ino, st.st_size) != image.fd_id:
                    raise OSError("descriptor {} no longer names the image's file".format(fd))
                a_lo = lo - lo % _GRAN
                win = mmap.mmap(fd, hi - a_lo, access=mmap.ACCESS_READ, offset=a_lo)
            except (OSError, ValueError):       # a closed or unusual descriptor: the whole-file mapping serves
                win = None
            if win is not None:
                try:
                    _parallel_copy(dst, np.frombuffer(win, dtype=np.uint8)[lo - a_lo:])
                finally:
                    del win                     # (unmapped here, inside the loop)
                return
        _parallel_copy(dst, image[lo:hi])
        return
    jobs, o = [], 0
    for part in image.pieces(lo, hi):
        for a in range(0, len(part), 4 << 20):          # at most 4 MiB per task
            b = min(len(part), a + (4 << 20))
            jobs.append((dst[o + a:o + b], part[a:b]))
        o += len(part)
    if len(jobs) < 2 or _COPY_THREADS == 1 or o < (8 << 20):
        for d, src in jobs:
            d[:] = src
        return
    if _copy_pool is None:
        _copy_pool = ThreadPoolExecutor(_COPY_THREADS)
    for f in [_copy_pool.submit(np.copyto, d, src) for d, src in jobs]:
        f.result()


# Pinned staging buffers are kept between pipelines: pinning 2 x 64 MiB costs
# 5.7-6 ms per reader (hipHostMalloc; torch's pinned allocator did not hand the
# blocks back in time for the next reader) -- 12 % of a 2 GiB read
# (profiles/r03i_prof_pipeline_windows.log).  At most `_PINNED_KEEP` bytes are
# kept; `release_pinned()` drops them.  THREE buffers per pipeline (two until late in
# round 4): with two, the page-cache copy of window k+1 can only start when the H2D
# of window k-1 has ended and has the length of one H2D (1.15 ms) to finish in; a
# copy that takes longer -- the host is shared -- idles the link.  cfg2 47.8-48.7 ->
# 52.9-53.0 GB/s, Mark 4 48.8-52.0 -> 52.4-53.1 on a box with slow copies
# (profiles/r04zs_pipeline_buffers.log); four: no further gain.
_NBUF = int(os.environ.get('BB_STAGING_BUFFERS', 0)) or 3      # pinned buffers per pipeline
_PINNED_KEEP = 512 << 20
_pinned_pool = []           # [(capacity, tensor)]


_pinned_lock = threading.Lock()     # (readers on several threads, the background uploader)


def _pinned_take(cap):
    with _pinned_lock:
        for i, (c, t) in enumerate(_pinned_pool):
            if c >= cap and c <= 2 * cap:
                del _pinned_pool[i]
                return t
    return torch.empty(cap, dtype=torch.uint8, pin_memory=True)


def _pinned_give(t):
    if t is None:
        return
    with _pinned_lock:
        if sum(c for c, _ in _pinned_pool) + t.numel() <= _PINNED_KEEP:
            _pinned_pool.append((t.numel(), t))


def release_pinned():
    """Drop the pinned staging buffers kept for the next reader."""
    with _pinned_lock:
        _pinned_pool.clear()


# When a list, WindowPipeline.run appends one record per window: host time
# waiting for a pinned buffer, host time of the page-cache copy, host time to
# enqueue, and the events around the H2D copy and around the window's kernels
# (`window_trace_summary`).  bench.py's `pipeline` leg and tools/ set it.
trace = None


def window_trace_summary(rows):
    """Totals of a `trace` list (call after the read has been synchronised)."""
    if not rows:
        return None
    nb = sum(r["bytes"] for r in rows)
    h2d = sum(r["events"][0].elapsed_time(r["events"][1]) for r in rows)
    ker = sum(r["events"][2].elapsed_time(r["events"][3]) for r in rows)
    host = {k: sum(r[k] for r in rows) for k in ("wait_ms", "host_copy_ms", "enqueue_ms", "enqueue_h2d_ms")}
    # the copy stream from the start of the first H2D to the end of the last one
    # (span - h2d_ms = time the link stood idle in between), what the compute
    # stream still had to do after the last copy, and the host's time before
    # the first copy was queued (first window's page-cache copy and set-up)
    span = rows[0]["events"][0].elapsed_time(rows[-1]["events"][1])
    tail = rows[-1]["events"][1].elapsed_time(rows[-1]["events"][3])
    lead = (rows[0]["t_end"] - rows[0]["t_start"]) * 1e3
    return {"windows": len(rows), "bytes": nb, "h2d_span_ms": round(span, 2), "kernels_after_last_h2d_ms": round(tail, 2),
            "host_before_first_h2d_ms": round(lead, 2),
            "host_wait_for_buffer_ms": round(host["wait_ms"], 2), "host_copy_ms": round(host["host_copy_ms"], 2),
            "host_enqueue_ms": round(host["enqueue_ms"], 2), "host_enqueue_h2d_ms": round(host["enqueue_h2d_ms"], 2),
            "host_copy_GBps": round(nb / max(host["host_copy_ms"], 1e-6) / 1e6, 1),
            "h2d_ms": round(h2d, 2), "h2d_GBps": round(nb / max(h2d, 1e-6) / 1e6, 1), "kernels_ms": round(ker, 2)}


class WindowPipeline:
    """Stream byte windows of a host image through pinned buffers to HBM and
    call ``process(dev_bytes, index)`` for each on the compute stream."""

    def __init__(self, image, max_window_bytes, nbuf=None, device='cuda'):
        if nbuf is None:
            nbuf = _NBUF
        self.image = image
        self.nbuf = nbuf
        self.device = torch.device(device)
        self.cap = int(max_window_bytes)
        self._pinned = [None] * nbuf
        self._dev = [None] * nbuf
        self._done = [None] * nbuf
        self._copy_stream = None
        self._count = 0

    def _buffers(self, b, need_dev=True):
        if self._pinned[b] is None:
            self._pinned[b] = _pinned_take(self.cap)
        if need_dev and self._dev[b] is None:
            # (windows that go to their own place in a file-sized `sink` need no
            # rotating device buffer: 2 x 64 MiB less to allocate per read,
            # profiles/r03h_prof_pipeline_windows.log)
            # +256 slack: kernels may read whole dwords at the tail
            self._dev[b] = torch.empty(self.cap + 256, dtype=torch.uint8, device=self.device)
            # the caching allocator may hand back a block whose previous owner
            # still has kernels queued on the compute stream: the side-stream
            # copy into it must come after them
            if self._copy_stream is not None:
                self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        return self._pinned[b], self._dev[b]

    def release(self):
        """Wait for the work queued on the buffers and drop them (torch's
        caching allocators hand the same pinned / device blocks to the next
        reader; keeping an own pool of them measured no gain, and 20 % slower
        64 MiB windows: `BB_STAGING_POOL` A/B in profiles/r01i_exp_staging_pool.log)."""
        self.drain()
        for t in self._pinned:
            _pinned_give(t)             # (drained: no copy out of them is pending)
        self._pinned = [None] * self.nbuf
        self._dev = [None] * self.nbuf
        self._done = [None] * self.nbuf

    def run(self, ranges, process, sink=None):
        """`sink`: optional device tensor as large as the image; windows are
        then copied to their own place in it (and stay there) instead of into
        the two rotating device buffers, and `process` gets that slice."""
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        for i, (lo, hi) in enumerate(ranges):
            n = hi - lo
            if n < 0:
                raise ValueError("window ends before it starts ({} > {})".format(lo, hi))
            if n > self.cap:
                raise ValueError("window larger than staging buffer")
            b = self._count % self.nbuf         # rotation continues across run() calls
            self._count += 1
            tr = trace                          # (per-window times for bench.py / tools: None in normal use)
            t0 = time.perf_counter() if tr is not None else 0.0
            if self._done[b] is not None:
                self._done[b].synchronize()          # buffer b free again
            t1 = time.perf_counter() if tr is not None else 0.0
            pinned, dev = self._buffers(b, need_dev=sink is None)
            target = dev[:n] if sink is None else sink[lo:hi]
            _stage(pinned.numpy(), self.image, lo, hi)           # page cache -> pinned (CPU)
            t2 = time.perf_counter() if tr is not None else 0.0
            with torch.cuda.stream(self._copy_stream):
                if tr is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self._copy_stream)
                target.copy_(pinned[:n], non_blocking=True)
                copied = torch.cuda.Event(enable_timing=tr is not None)
                copied.record(self._copy_stream)
            main.wait_event(copied)
            if tr is not None:
                t2b = time.perf_counter()
                k0 = torch.cuda.Event(enable_timing=True)
                k0.record(main)
            process(target, i)
            done = torch.cuda.Event(enable_timing=tr is not None)
            done.record(main)
            self._done[b] = done
            if tr is not None:
                t3 = time.perf_counter()
                tr.append({"bytes": n, "wait_ms": (t1 - t0) * 1e3, "host_copy_ms": (t2 - t1) * 1e3,
                           "enqueue_ms": (t3 - t2) * 1e3, "enqueue_h2d_ms": (t2b - t2) * 1e3, "t_start": t0, "t_end": t3, "events": (e0, copied, k0, done)})

    def drain(self):
        for ev in self._done:
            if ev is not None:
                ev.synchronize()


_SMALL_BYTES = 2 << 20
_small_ring = []            # [[pinned tensor, event of the copy that last read it]]
_small_next = 0
_small_lock = threading.Lock()


def _small_slot():
    """The next pinned scratch buffer of the ring, free to be written (the
    copy that read it last has finished)."""
    global _small_next
    if not _small_ring:
        for _ in range(4):
            _small_ring.append([torch.empty(_SMALL_BYTES + 256, dtype=torch.uint8, pin_memory=True), None])
    slot = _small_ring[_small_next]
    _small_next = (_small_next + 1) % len(_small_ring)
    if slot[1] is not None:
        slot[1].synchronize()
    return slot


def upload(image, device='cuda', chunk_bytes=64 << 20, join=True):
    """Whole host image -> one device tensor (with 256 bytes of slack), moved
    in pinned chunks on a side stream while the next chunk is being copied by
    the CPU.  Used when a file is kept resident in HBM.  With ``join=False``
    the caller's stream is not made to wait for the copy: returns ``(tensor,
    event)`` and whoever uses the tensor waits for the event first
    (`upload_in_background`)."""
    device = torch.device(device)
    n = len(image)
    dev = empty_output((n + 256,), dtype=torch.uint8, device=device, create=False)
    if n <= _SMALL_BYTES:
        # small windows (random access): through one of a few pinned scratch
        # buffers, zero tail included, with ONE asynchronous copy (a pageable
        # source makes the copy synchronous: 28 us for two frames; the tail's
        # own fill was another launch)
        with _small_lock:
            slot = _small_slot()
            host = slot[0].numpy()
            if n:
                host[:n] = image[:n]
            host[n:n + 256] = 0
            dev.copy_(slot[0][:n + 256], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(device))
        if not join:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            return dev, ev
        return dev
    dev[n:] = 0
    stream = torch.cuda.Stream(device=device)
    # `dev` may be a recycled block with work of its previous owner still
    # queued on the compute stream (and the tail was just zeroed there)
    stream.wait_stream(torch.cuda.current_stream(device))
    # (from the pool of pinned buffers kept between readers: pinning 2 x 64 MiB
    # costs 6-25 ms, which a loop of small reads paid once per reader while its
    # read-ahead windows grew)
    cap = chunk_bytes               # (one size: every later window finds it in the pool)
    n_second = 0 if n <= chunk_bytes else cap
    pinned = [_pinned_take(cap), _pinned_take(n_second) if n_second else None]
    events = [None, None]
    for i, lo in enumerate(range(0, n, chunk_bytes)):
        hi = min(n, lo + chunk_bytes)
        b = i % 2
        if events[b] is not None:
            events[b].synchronize()
        _stage(pinned[b].numpy(), image, lo, hi)
        with torch.cuda.stream(stream):
            dev[lo:hi].copy_(pinned[b][:hi - lo], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        events[b] = ev
    if join:
        torch.cuda.current_stream(device).wait_stream(stream)
    for ev in events:
        if ev is not None:
            ev.synchronize()
    for t in pinned:
        _pinned_give(t)
    if not join:
        done = torch.cuda.Event()
        done.record(stream)
        return dev, done
    return dev


_prefetch_pool = None


def upload_in_background(image, lo, hi, device=None):
    """Start moving bytes [lo, hi) of a host image to HBM on a worker thread
    (page cache -> pinned chunks -> hipMemcpyAsync on a side stream) and return
    a future of ``(device tensor, event)``: the next block of a sequential
    reader travels while the caller is still working on the current one --
    the overlap `WindowPipeline` gives inside one large read, carried across
    `read()` calls.  The consumer makes its stream wait for the event."""
    global _prefetch_pool
    if _prefetch_pool is None:
        _prefetch_pool = ThreadPoolExecutor(1, thread_name_prefix='bb-prefetch')
    index = torch.cuda.current_device() if device is None else torch.device(device).index

    def job():
        with torch.cuda.device(index):
            return upload(image[lo:hi], device=torch.device('cuda', index), join=False)
    return _prefetch_pool.submit(job)


def download(dev, host, chunk_bytes=None):
    """Device tensor -> NumPy array of the same shape and dtype, through two
    pinned buffers: the DMA engine fills one while the copy threads empty the
    other into `host` (a pageable destination copied in one go runs at about a
    third of the link rate).  Used by ``read(out=<ndarray>)`` for large reads.
    Pieces of an eighth of the array (4 to 64 MiB), so that a result of a few
    hundred MiB overlaps its two halves as well."""
    flat = dev.contiguous().reshape(-1)
    if flat.is_complex():
        flat = torch.view_as_real(flat).reshape(-1)
    src = flat.view(torch.uint8)
    dst = host.reshape(-1).view(np.uint8)
    n = src.numel()
    assert dst.size == n and host.flags.c_contiguous
    if n == 0:
        return host
    if chunk_bytes is None:
        chunk_bytes = min(64 << 20, max(4 << 20, (n // 8 + 4095) & ~4095))
    stream = torch.cuda.Stream(device=dev.device)
    stream.wait_stream(torch.cuda.current_stream(dev.device))
    pinned = [torch.empty(min(chunk_bytes, n), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    events = [None, None]
    spans = [(lo, min(n, lo + chunk_bytes)) for lo in range(0, n, chunk_bytes)]
    for i in range(len(spans) + 1):
        if i < len(spans):
            lo, hi = spans[i]
            b = i % 2
            with torch.cuda.stream(stream):
                pinned[b][:hi - lo].copy_(src[lo:hi], non_blocking=True)
                events[b] = torch.cuda.Event()
                events[b].record(stream)
        if i > 0:                                   # drain chunk i-1 while chunk i is in flight
            plo, phi = spans[i - 1]
            pb = (i - 1) % 2
            events[pb].synchronize()
            _parallel_copy(dst[plo:phi], pinned[pb].numpy()[:phi - plo])
    return host


# ``read()`` results handed to a caller of the reference are NEW host arrays.  Up to
# `_PINNED_RESULT_MAX` bytes each -- the chunked loops of user scripts -- they are
# arrays ON pinned memory (a pinned torch tensor seen as ndarray): the DMA engine
# writes the result where the caller reads it and no core copies anything; the
# block goes back to torch's pinned allocator when the array is dropped, so a loop
# alternates between two blocks.  (Through pinned staging and a copy into fresh
# pageable memory a 64 MiB read ran at 11 GB/s: 1.2 ms of DMA, then 4 ms of page
# faults and copying, nothing overlapping; profiles/r05zz_dropin_read.log.)
# Larger results, and results beyond `_PINNED_RESULT_TOTAL` bytes outstanding
# (a script that collects everything it reads), take `download`.
_PINNED_RESULT_MIN = 1 << 20
_PINNED_RESULT_MAX = int(os.environ.get('BB_PINNED_RESULT_MAX', 1 << 30))
_PINNED_RESULT_TOTAL = int(os.environ.get('BB_PINNED_RESULT_TOTAL', 8 << 30))
_pinned_results = {'bytes': 0, 'made': 0}
_pinned_results_lock = threading.Lock()


def _pinned_result_dropped(nbytes):
    with _pinned_results_lock:
        _pinned_results['bytes'] -= nbytes


def download_new(dev):
    """A new NumPy array with the device tensor's contents (float32 or
    complex64, same shape)."""
    nbytes = dev.numel() * dev.element_size()
    if nbytes < _PINNED_RESULT_MIN or not dev.is_cuda:
        return dev.cpu().numpy()
    take = False
    if nbytes <= _PINNED_RESULT_MAX:
        with _pinned_results_lock:
            if _pinned_results['bytes'] + nbytes <= _PINNED_RESULT_TOTAL:
                _pinned_results['bytes'] += nbytes
                _pinned_results['made'] += 1
                take = True
    if take:
        try:
            host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        except RuntimeError:                        # no pinned memory to be had: the staged copy
            host = None
        if host is None:
            _pinned_result_dropped(nbytes)
        else:
            host.copy_(dev, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev.device))
            done.synchronize()
            arr = host.numpy()                      # (keeps `host` alive; views of `arr` keep `arr` alive)
            weakref.finalize(arr, _pinned_result_dropped, nbytes)
            return arr
    out = np.empty(tuple(dev.shape), dtype=np.complex64 if dev.is_complex() else np.float32)
    return download(dev, out)


class _FileSink:
    """Ordered background writes to ONE file handle: the caller's thread queues
    device-to-pinned copies (side stream, events) and host bytes; a thread of
    the sink waits for each copy and puts the bytes into the file -- so the
    ``write()`` of piece k overlaps the encode and the device-to-host copy of
    everything behind it, ACROSS the writer's ``write()`` calls (round 5: the
    buffered write of a piece was 80 % of a call and ran alone,
    profiles/r04zn_prof_writer.log).  At most `depth` pinned pieces are in
    flight.  An error of the file (a full disk) is raised by the next call
    that touches the sink.

    A handle that takes POSITIONAL writes (`helpers.sequentialfile.
    SequentialFileWriter.can_pwrite`: a sequence of files of a fixed size) gets
    `_NWORKER` threads instead of one: every piece carries its offset in the
    stream and goes to the thread of the FILE it starts in (file number modulo
    the threads), so that several files fill at the same time -- one file takes
    11-12 GB/s from any number of threads, 2 / 4 / 8 files at once 23 / 40 / 69
    (profiles/r05g_exp_file_write2.log) -- with `depth` = 16 pieces (512 MiB) in
    flight, enough for four 128 MiB frames."""
    _NWORKER = int(os.environ.get('BB_WRITE_THREADS', 4))

    def __init__(self, fh, depth=None):
        # the handle the registry knows this sink by.  Its id is reused once it is freed, so a
        # lookup checks identity; held WEAKLY so that a writer dropped without close() frees it
        # and `_sink_for`'s finalizer then settles and closes this sink
        try:
            self.key = weakref.ref(fh)
        except TypeError:               # a handle that cannot be weakly referenced is simply kept
            self.key = lambda fh=fh: fh
        while getattr(type(fh), '_queues_writes', False):
            fh = fh.fh_raw              # (a FileBase settles its queue before it answers: write below it)
        self.fh = fh
        self.positional = bool(_WRITE_ASYNC and getattr(fh, 'can_pwrite', False) and self._NWORKER > 1)
        nq = self._NWORKER if self.positional else 1
        self.depth = depth or (16 if self.positional else 4)
        self.qs = [queue.Queue() for _ in range(nq)]
        self.slots = threading.Semaphore(self.depth)
        self.error = None
        self.stream = None
        self.pos = fh.tell() if self.positional else 0      # stream offset of the next byte queued
        self._settled = False           # drained and nothing queued since: the handle's own position leads
        self.closed = False
        self.threads = [threading.Thread(target=self._run, args=(q,), name='bb-file-sink', daemon=True)
                        for q in self.qs]
        for t in self.threads:
            t.start()

    def _run(self, q):
        while True:
            item = q.get()
            try:
                if item is None:
                    return
                kind, payload, n, event, offset = item
                if kind == 'dev':
                    # ALWAYS, also after a file error: the buffer goes back to the shared
                    # pinned pool below and its device-to-host copy must have landed first
                    try:
                        event.synchronize()
                    except BaseException as exc:
                        payload = None                  # state of the copy unknown: never pooled again
                        if self.error is None:
                            self.error = exc
                if self.error is None:
                    try:
                        if kind == 'dev':
                            data = memoryview(payload.numpy()[:n])
                        else:
                            data = payload
                        if offset is None:
                            self.fh.write(data)
                        else:
                            self.fh.pwrite_stream(data, offset)
                    except BaseException as exc:        # kept for the caller's thread
                        self.error = exc
                if kind == 'dev':
                    if payload is not None:
                        _pinned_give(payload)
                    self.slots.release()
            finally:
                q.task_done()

    def _check(self):
        if self.error is not None:
            exc, self.error = self.error, None
            raise exc

    def _queue_for(self, n):
        """(queue, stream offset or None) for the next `n` bytes."""
        if not self.positional:
            return self.qs[0], None
        if self._settled:
            # nothing in flight: bytes written on the handle itself since the last drain
            # (``fw.fh_raw.write_frame(...)`` between stream writes) moved it past `pos`
            self.pos = max(self.pos, self.fh.tell())
            self._settled = False
        offset = self.pos
        self.pos += n
        return self.qs[(offset // self.fh.file_size) % len(self.qs)], offset

    def put_device(self, src, chunk_bytes):
        self._check()
        dev = src.device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        n = src.numel()
        npiece = max(1, (n + chunk_bytes // 4) // chunk_bytes)          # (no small tail piece: up to 1.25 chunks go as one)
        step = -(-n // npiece)
        step += -step % 4096
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            self.slots.acquire()                        # back-pressure: `depth` pieces in flight
            host = _pinned_take(step)
            with torch.cuda.stream(self.stream):
                host[:hi - lo].copy_(src[lo:hi], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            q, offset = self._queue_for(hi - lo)
            q.put(('dev', host, hi - lo, ev, offset))
        # (the tensor must outlive the copies that read it on the side stream)
        try:
            src.record_stream(self.stream)
        except Exception:               # memory torch's allocator does not manage (an arena block)
            torch.cuda.current_stream(dev).wait_stream(self.stream)

    def put_host(self, data):
        self._check()
        data = bytes(data)
        q, offset = self._queue_for(len(data))
        q.put(('host', data, 0, None, offset))

    def drain(self):
        for q in self.qs:
            q.join()
        if self.positional:
            self.fh.sync_position(self.pos)
            self._settled = True
        self._check()

    def close(self):
        if self.closed:
            return
        self.closed = True
        for q in self.qs:
            q.join()
        for q in self.qs:
            q.put(None)
        for t in self.threads:
            t.join()
        if self.positional and self.error is None:
            self.fh.sync_position(self.pos)
        self._check()


_sinks = {}                 # id(file handle) -> _FileSink (which keeps the handle: `_FileSink.key`)
_sinks_lock = threading.Lock()
_WRITE_ASYNC = os.environ.get('BB_WRITE_ASYNC', '1') not in ('0', 'no', 'off')


def _drain_all_sinks():
    """At interpreter exit: what writers that were never closed have queued still
    reaches their files (the sinks' threads are daemons and would die with it)."""
    with _sinks_lock:
        sinks = list(_sinks.values())
        _sinks.clear()
    for s_ in sinks:
        try:
            s_.close()
        except BaseException:
            pass


atexit.register(_drain_all_sinks)


def _sink_for(fh, create=True):
    with _sinks_lock:
        s = _sinks.get(id(fh))
        if s is not None and s.key() is not fh:         # the sink of a dead handle whose id `fh` now has
            s = None
        if s is None and create:
            s = _sinks[id(fh)] = _FileSink(fh)
            try:
                weakref.finalize(fh, _drop_sink, id(fh), s)
            except TypeError:
                pass
        return s


def _drop_sink(key_id, sink):
    """The handle `sink` was made for is gone without `finish_writes`: what it queued still
    reaches the file, its thread ends and its pinned pieces go back to the pool."""
    with _sinks_lock:
        if _sinks.get(key_id) is sink:
            del _sinks[key_id]
        elif sink.closed:
            return
    try:
        sink.close()
    except BaseException:               # (a finalizer has nobody to raise to)
        pass


def write_host_bytes(fh, data):
    """Host bytes -> `fh`, IN ORDER with the device pieces `write_device_bytes`
    has queued for the same handle (frame headers of the block formats)."""
    s = _sink_for(fh, create=False)
    if s is None:
        fh.write(data)
    else:
        s.put_host(data)


class HostWriteOrder:
    """File-like ``write`` for header ``tofile`` calls that must stay in order
    with queued device pieces: ``header.tofile(HostWriteOrder(fh))``."""

    def __init__(self, fh):
        self.fh = fh

    def write(self, data):
        write_host_bytes(self.fh, data)
        return len(data)


def finish_writes(fh, close_sink=True):
    """Wait until everything queued for `fh` is in the file (before the handle
    is closed, flushed, or asked where it stands); raises what a write raised."""
    with _sinks_lock:
        s = _sinks.get(id(fh))
        if s is not None and s.key() is not fh:
            s = None
        elif s is not None and close_sink:
            del _sinks[id(fh)]
    if s is not None:
        s.close() if close_sink else s.drain()


def write_device_bytes(fh, dev, chunk_bytes=32 << 20):
    """Device uint8 tensor -> `fh` at its current position (the stream
    writers' way to the file; the reference fills a memory map of the file
    frame by frame, base/base.py:1276-1342).  ASYNCHRONOUS since round 5: the
    pieces (32 MiB) travel to pinned buffers on a side stream and a thread of
    the handle's `_FileSink` writes them in order while the caller encodes the
    next frames; `finish_writes(fh)` -- the writers call it when they close --
    waits for them.  BB_WRITE_ASYNC=0: each call returns when its bytes are in
    the file.  One ``write`` at a time is all a file takes: on the GPU box 12
    GB/s into a new file whatever the number of threads (buffered writes to
    one file serialise on its inode lock: profiles/r03y_exp_file_write.log);
    files written AT THE SAME TIME scale -- 2 / 4 / 8 files 23 / 40 / 69 GB/s
    (profiles/r05g_exp_file_write2.log) -- which the GSB writer's several raw
    files get from their own sinks."""
    src = dev.reshape(-1)
    if src.numel() == 0:
        return
    s = _sink_for(fh)
    s.put_device(src, chunk_bytes)
    if not _WRITE_ASYNC:
        s.drain()


def to_numpy(dev):
    """Device tensor -> new NumPy array: `download` for large tensors, a plain
    copy for small ones."""
    if dev.is_cuda and dev.numel() * dev.element_size() >= (32 << 20):
        out = np.empty(tuple(dev.shape), dtype=np.complex64 if dev.is_complex() else
                       {torch.float32: np.float32, torch.uint8: np.uint8}.get(dev.dtype, None))
        if out.dtype in (np.float32, np.complex64):
            return download(dev, out)
    return dev.cpu().numpy()


def upload_array(array, device='cuda'):
    """float32 / complex64 NumPy array -> device tensor of the same shape:
    large C-contiguous arrays go through the pinned chunk pipeline of `upload`
    (a pageable source copied in one go runs at a fifth of the link rate)."""
    array = np.asarray(array)
    tdtype = {np.dtype(np.float32): torch.float32, np.dtype(np.complex64): torch.complex64}.get(array.dtype)
    if tdtype is None or not array.flags.c_contiguous or array.nbytes < (32 << 20):
        return torch.from_numpy(np.ascontiguousarray(array)).to(device)
    raw = array.reshape(-1).view(np.uint8)
    dev = upload(raw, device=device)[:raw.size]
    return dev.view(tdtype).reshape(array.shape)
