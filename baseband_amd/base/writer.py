"""Stream writer base: samples in (device tensors or arrays), frames out.

Write-side twin of `GPUStreamReaderBase` with the
call shape of the reference's ``StreamWriterBase`` (base/base.py:1230-1342):
``write(data, valid=True)`` buffers until whole frames are available, the
samples of the complete frames are packed by the GPU encoders
(``bb_encode_flat`` / ``bb_encode_mark4``), and the host only glues headers
and payload bytes together.  ``close()`` pads a partial last frame with zeros
and marks it invalid, with the reference's warning.
"""
import warnings

import numpy as np
import torch
from .quantities import hz


class LazyWriteFile:
    """File that is created (truncated) only when the first bytes are written
    or on close, so that a writer whose arguments turn out to be invalid never
    clobbers an existing file."""

    def __init__(self, name):
        self.name = name
        self._fh = None

    mode = 'rb+'                        # what io.open(name, 'w+b') reports

    def _open(self):
        if self._fh is None:
            self._fh = open(self.name, 'w+b')
        return self._fh

    # frames a stream writer has packed are queued for this handle and written behind
    # its back (staging.write_device_bytes -- to the file itself, `fh_raw`): whoever asks
    # where the file stands, or writes or seeks on the side, finds every frame in place
    _queues_writes = True
    fh_raw = property(lambda self: self._open())

    def _settle(self):
        from .. import staging
        if staging._sinks:
            staging.finish_writes(self, close_sink=False)

    def write(self, data):
        self._settle()
        return self._open().write(data)

    def tell(self):
        self._settle()
        return self._fh.tell() if self._fh is not None else 0

    def seek(self, offset, whence=0):
        self._settle()
        return self._open().seek(offset, whence)

    def read(self, count=-1):
        return self._open().read(count)

    def flush(self):
        if self._fh is not None:
            self._fh.flush()

    def fileno(self):
        return self._open().fileno()

    def readable(self):
        return True

    def writable(self):
        return True

    def seekable(self):
        return True

    @property
    def closed(self):
        return self._fh is not None and self._fh.closed

    def memmap(self, dtype=np.uint8, mode=None, offset=None, shape=None, order='C'):
        """Writable map of the next bytes (the file grows to hold them), as
        ``numpy.memmap`` on a file opened 'w+b' gives the reference's
        ``memmap_frame`` (dada/base.py:185-208)."""
        if shape is None:
            raise ValueError('cannot make writable memmap without shape.')
        fh = self._open()
        start = fh.tell() if offset is None else offset
        mm = np.memmap(fh, dtype, mode or 'r+', start, shape, order)
        fh.seek(start + mm.nbytes)
        return mm

    def close(self):
        self._open().close()

    def discard(self):
        """Give up without creating the file (the writer never came to be)."""
        if self._fh is not None:
            self._fh.close()


class GPUStreamWriterBase:
    def __init__(self, fh_raw, header0, *, sample_rate, samples_per_frame,
                 unsliced_shape, bps, complex_data, squeeze=True):
        self.fh_raw = fh_raw
        self.header0 = header0
        self.sample_rate = hz(sample_rate)
        self.samples_per_frame = samples_per_frame
        self._unsliced_shape = tuple(unsliced_shape)
        self.bps = bps
        self.complex_data = complex_data
        self.squeeze = bool(squeeze)
        self.offset = 0                     # samples accepted so far
        self._nframes_written = 0
        self._pending = []                  # list of (tensor, valid)
        self._npending = 0
        self._closed = False

    @property
    def sample_shape(self):
        key = (self.squeeze, tuple(self._unsliced_shape))
        cached = self.__dict__.get('_sample_shape_cached')
        if cached is None or cached[0] != key:
            from .utils import named_sample_shape
            fields = self._sample_shape_fields
            if callable(fields):
                fields = fields(len(self._unsliced_shape))
            cached = self._sample_shape_cached = (key, named_sample_shape(self._unsliced_shape, fields, self.squeeze))
        return cached[1]

    _sample_shape_fields = None

    def __reduce__(self):
        raise TypeError('cannot pickle file opened for writing')

    def readable(self):
        """A stream writer cannot be read from (base/base.py:559-567 in the reference)."""
        return False

    def writable(self):
        return not self._closed

    @property
    def closed(self):
        return self._closed

    def seekable(self):
        return False

    def __repr__(self):
        def attr(name):
            try:
                return getattr(self, name)
            except Exception:
                return None
        return ("<{cls} name={name} offset={offset}\n"
                "    sample_rate={rate}, samples_per_frame={spf},\n"
                "    sample_shape={shape}, bps={bps},\n"
                "    start_time={start}>"
                .format(cls=type(self).__name__, name=getattr(self.fh_raw, 'name', None), offset=attr('offset'),
                        rate=attr('sample_rate'), spf=attr('samples_per_frame'), shape=attr('sample_shape'),
                        bps=attr('bps'), start=attr('start_time')))

    @property
    def start_time(self):
        return self._start_time

    @property
    def time(self):
        """Time of the next sample to be written."""
        return self.tell('time')

    def tell(self, unit=None):
        if unit is None:
            return self.offset
        ns = int(round(self.offset * 1e9 / self.sample_rate))
        if unit == 'time':
            return self._start_time + np.timedelta64(ns, 'ns')
        if unit == 's':
            return self.offset / self.sample_rate
        if hasattr(unit, 'to'):
            # a unit object of the caller's (astropy: ``fw.tell(u.us)``), as the readers' tell does
            return (self.offset / self.sample_rate / float(unit.to('s'))) * unit
        raise ValueError("unit should be None, 'time', 's' or a unit of time")

    def write(self, data, valid=True):
        """Accept `data` of shape ``(n,) + sample_shape`` (NumPy array or torch
        tensor on any device)."""
        if self._closed:
            raise ValueError("I/O operation on closed stream.")
        if not isinstance(data, torch.Tensor):
            from ..staging import upload_array
            data = upload_array(data)
        assert tuple(data.shape[1:]) == self.sample_shape, (
            "'data' should have trailing shape {}".format(self.sample_shape))
        want = torch.complex64 if self.complex_data else torch.float32
        data = data.to(device='cuda', dtype=want).reshape(
            (data.shape[0],) + self._unsliced_shape)
        self._pending.append((data, bool(valid)))
        self._npending += data.shape[0]
        self.offset += data.shape[0]
        spf = self.samples_per_frame
        nfull = self._npending // spf
        if nfull:
            block = torch.cat([d for d, _ in self._pending]) if len(self._pending) > 1 \
                else self._pending[0][0]
            # validity per frame: a frame is valid only if all of its pieces were
            edges = np.cumsum([0] + [d.shape[0] for d, _ in self._pending])
            flags = np.ones(nfull, bool)
            for (lo, hi), (_, ok) in zip(zip(edges[:-1], edges[1:]), self._pending):
                if not ok and hi > lo:
                    flags[lo // spf:min(nfull, -(-hi // spf))] = False
            self._write_frames(block[:nfull * spf], flags)
            self._nframes_written += nfull
            rest = block[nfull * spf:]
            tail_ok = self._pending[-1][1]
            self._pending = [(rest, tail_ok)] if rest.shape[0] else []
            self._npending = rest.shape[0]

    def _write_frames(self, data, valid):
        raise NotImplementedError

    def _emit_frames(self, header_bytes, packed, fh=None):
        """Glue ``header_bytes`` (host uint8 array, one row per frame) in front
        of the payload rows of ``packed`` (device uint8 tensor) ON THE GPU,
        bring whole frames to the host with one copy into a pinned buffer and
        hand that buffer to the file -- no per-frame work and no further host
        copies (three of them, at 5-10 GB/s each, used to sit between the
        encoder and the file; tools/bench_writers.py).  Large writes are
        pipelined, copy against write (`staging.write_device_bytes`)."""
        nfr, hn = header_bytes.shape
        pk = packed.reshape(nfr, -1)
        frames = torch.empty((nfr, hn + pk.shape[1]), dtype=torch.uint8, device=pk.device)
        frames[:, hn:] = pk
        frames[:, :hn] = torch.from_numpy(np.ascontiguousarray(header_bytes)).to(pk.device)
        from ..staging import write_device_bytes
        write_device_bytes(self.fh_raw if fh is None else fh, frames)

    def _to_host(self, dev):
        """Device uint8 tensor -> flat uint8 NumPy view of this writer's pinned
        buffer (valid until the next call)."""
        n = dev.numel()
        if self._staging is None or self._staging.numel() < n:
            self._staging = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        host = self._staging[:n]
        host.copy_(dev.reshape(-1))
        return host.numpy()

    def close(self):
        if self._closed:
            return
        extra = self._npending
        if extra:
            warnings.warn("closing with partial buffer remaining.  "
                          "Writing padded frame, marked as invalid.")
            pad = torch.zeros((self.samples_per_frame - extra,) + self._unsliced_shape,
                              dtype=self._pending[0][0].dtype, device='cuda')
            self._pending.append((pad, False))
            self._npending += pad.shape[0]
            block = torch.cat([d for d, _ in self._pending])
            self._write_frames(block, np.zeros(1, bool))
            self._nframes_written += 1
            self._pending, self._npending = [], 0
        self._closed = True
        try:
            self._finish_writes()
        finally:
            self._close_files()

    def _raw_files(self):
        """Every handle device bytes are written to (flattened)."""
        out, todo = [], [self.fh_raw]
        while todo:
            x = todo.pop()
            if isinstance(x, (tuple, list)):
                todo.extend(x)
            elif x is not None:
                out.append(x)
        return out

    def flush(self):
        """Wait until everything written so far is in the file(s) (the pieces
        travel through a background sink: staging.write_device_bytes)."""
        from ..staging import finish_writes
        for fh in self._raw_files():
            finish_writes(fh, close_sink=False)
            if hasattr(fh, 'flush'):
                fh.flush()

    def _finish_writes(self):
        from ..staging import finish_writes
        err = None
        for fh in self._raw_files():
            try:
                finish_writes(fh)
            except BaseException as exc:
                err = err or exc
        if err is not None:
            raise err

    def _close_files(self):
        self.fh_raw.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BlockStreamWriter(GPUStreamWriterBase):
    """Writer for block formats (DADA, GUPPI): every frame is an ASCII header
    followed by int8 samples in the format's storage order.  Samples are
    rounded, clipped and packed on the GPU (``bb_encode_flat``, integer
    coder); subclasses give the per-frame header (`_frame_header`) and the
    storage order (`_storage_order`).

    Replaces the memory-mapped frame filling of the reference's writers
    (dada/base.py:333-362, guppi/base.py:281-310, base/base.py:1276-1342); a
    partial last frame is padded with zeros at close, as there."""

    def _frame_header(self, index):
        raise NotImplementedError

    def _storage_order(self, block):
        """(nframes, samples_per_frame, *sample_shape[, 2]) float32 tensor ->
        same values in on-disk order."""
        return block

    def _write_frames(self, data, valid):
        from .. import kernels, _lib
        spf = self.samples_per_frame
        nframes = data.shape[0] // spf
        if data.is_complex():
            data = torch.view_as_real(data)
        block = self._storage_order(data.reshape((nframes, spf) + tuple(data.shape[1:])))
        from ..staging import write_device_bytes, HostWriteOrder
        payloads = kernels.encode_flat(block, _lib.CODER_INT, self.bps).reshape(nframes, -1)
        for k in range(nframes):
            header = self._frame_header(self._nframes_written + k)
            assert header.payload_nbytes == payloads.shape[1]
            header.tofile(HostWriteOrder(self.fh_raw))       # (in order with the queued payloads)
            write_device_bytes(self.fh_raw, payloads[k])
