"""GSB stream reader and ``open`` (gsb/base.py:146-387).

GSB data are headerless fixed-size blocks in one (rawdump) or several
(phased: polarisations x parts) raw files, with one line per block in a
separate timestamp file.  Frame k lives at ``k * payload_nbytes`` in every
raw file.  The phased layout -- parts consecutive in time, polarisations
interleaved per sample -- is exactly the multi-slot layout of
``bb_decode_frames``: output "frame" (k, part) with one slot per
polarisation.
"""
import io

import numpy as np
import torch

from .. import _lib, kernels
from ..base.base import FileBase, GPUStreamReaderBase
from ..base.writer import GPUStreamWriterBase, LazyWriteFile
from ..staging import host_image, retire_image, write_device_bytes
from .header import GSBHeader
from .payload import GSBPayload
from ..base.quantities import hz

__all__ = ['GSBTimeStampIO', 'GSBFileReader', 'GSBFileWriter', 'GSBStreamReader',
           'GSBStreamWriter', 'open']

DEFAULT_FRAME_RATE = 1e8 / 6 / 2 ** 22          # Hz (gsb/base.py:170)



def _raw_handles(fh_raw, rawdump):
    """Binary handles in the shape the stream classes use: one handle for
    rawdump, a rectangular (polarisation, file) nest for phased data
    (gsb/base.py:253-262, 403-411)."""
    if not isinstance(fh_raw, (tuple, list)):
        return fh_raw if rawdump else ((fh_raw,),)
    assert not rawdump, "rawdump data come in one file"
    nfile = len(fh_raw[0])
    assert all(isinstance(row, (tuple, list)) and len(row) == nfile for row in fh_raw)
    return fh_raw


class GSBTimeStampIO(FileBase):
    """Timestamp (text) file: one header line per frame (gsb/base.py:23-75)."""

    def read_timestamp(self):
        return GSBHeader.fromfile(self.fh_raw)

    def write_timestamp(self, header=None, **kwargs):
        if header is None:
            header = GSBHeader.fromvalues(**kwargs)
        return header.tofile(self.fh_raw)

    def get_frame_rate(self):
        """From the first two timestamps (Hz)."""
        with self.temporary_offset(0):
            t0 = self.read_timestamp().time
            t1 = self.read_timestamp().time
        return 1e9 / float((t1 - t0) / np.timedelta64(1, 'ns'))

    @property
    def info(self):
        """What the timestamp file says by itself (gsb/file_info.py:16-95)."""
        from .info import GSBTimeStampInfo
        return GSBTimeStampInfo(self)


class GSBFileReader(FileBase):
    """Raw data file: fixed-size payload blocks (gsb/base.py:78-121)."""

    def __init__(self, fh_raw, payload_nbytes, nchan=1, bps=4, complex_data=False):
        self.payload_nbytes, self.nchan = payload_nbytes, nchan
        self.bps, self.complex_data = bps, complex_data
        super().__init__(fh_raw)

    def __repr__(self):
        return ("{name}(fh_raw={s.fh_raw}, payload_nbytes={s.payload_nbytes}, nchan={s.nchan}, bps={s.bps}, "
                "complex_data={s.complex_data})".format(name=type(self).__name__, s=self))

    def read_payload(self):
        return GSBPayload.fromfile(self.fh_raw, payload_nbytes=self.payload_nbytes,
                                   sample_shape=(self.nchan,), bps=self.bps,
                                   complex_data=self.complex_data)


class GSBFileWriter(FileBase):
    """Raw data file writer (gsb/base.py:124-143): payloads packed on the GPU."""

    def write_payload(self, data, bps=4):
        if not isinstance(data, GSBPayload):
            data = GSBPayload.fromdata(data, bps=bps)
        return data.tofile(self.fh_raw)


def _stream_repr(self):
    """The reference's layout (gsb/base.py:240-254): timestamp file, data file(s), shape."""
    raw = self.fh_raw
    if isinstance(raw, (list, tuple)):
        data_name = tuple(tuple(str(getattr(p, 'name', None)).split('/')[-1] for p in pol) for pol in raw)
    else:
        data_name = getattr(raw, 'name', None)
    subset = getattr(self, 'subset', None)
    return ("<{cls} header={ts} offset= {s.offset}\n    data={dn}\n"
            "    sample_rate={s.sample_rate:.5g}, samples_per_frame={s.samples_per_frame},\n"
            "    sample_shape={s.sample_shape}, bps={s.bps},\n"
            "    {sub}start_time={s.start_time}>"
            .format(cls=type(self).__name__, s=self, ts=getattr(self.fh_ts, 'name', None), dn=data_name,
                    sub='subset={0}, '.format(subset) if subset else ''))


class GSBStreamReader(GPUStreamReaderBase):
    _sample_shape_fields = staticmethod(lambda n: ('nchan',) if n == 1 else ('nthread', 'nchan'))
    __repr__ = _stream_repr

    def __init__(self, fh_ts, fh_raw, sample_rate=None, samples_per_frame=None,
                 payload_nbytes=None, nchan=None, bps=None, complex_data=None,
                 squeeze=True, subset=(), verify=True):
        self.fh_ts = fh_ts = fh_ts if isinstance(fh_ts, GSBTimeStampIO) else GSBTimeStampIO(fh_ts)
        sample_rate = hz(sample_rate)           # (a Quantity from callers of the reference: base/quantities.py)
        lines = [ln for ln in fh_ts.read().splitlines() if ln.strip()]
        lines = [ln.decode('ascii') if isinstance(ln, bytes) else ln for ln in lines]
        header0 = GSBHeader(lines[0].split())
        rawdump = header0.mode == 'rawdump'
        fh_raw = _raw_handles(fh_raw, rawdump)
        complex_data = (not rawdump) if complex_data is None else complex_data
        bps = bps if bps is not None else (4 if rawdump else 8)
        nchan = nchan if nchan is not None else (1 if rawdump else 512)
        bpfs = bps * nchan * (2 if complex_data else 1)
        nfiles = 1 if rawdump else len(fh_raw[0])
        if payload_nbytes is None:
            if samples_per_frame is None:
                if sample_rate is None:
                    payload_nbytes = 2 ** 22
                else:
                    payload_nbytes = int(round(sample_rate / DEFAULT_FRAME_RATE
                                               * bpfs / 8 / nfiles))
            else:
                payload_nbytes = samples_per_frame * bpfs // (8 * nfiles)
        if samples_per_frame is None:
            samples_per_frame = payload_nbytes * 8 // bpfs * nfiles
        elif samples_per_frame != payload_nbytes * nfiles * 8 / bpfs:
            raise ValueError('inconsistent samples_per_frame, bps, '
                             'complex_data, and payload_nbytes')
        if sample_rate is None:
            sample_rate = samples_per_frame * DEFAULT_FRAME_RATE
        shape = (nchan,) if rawdump else (len(fh_raw), nchan)
        super().__init__(
            fh_raw, header0, sample_rate=hz(sample_rate),
            samples_per_frame=samples_per_frame, unsliced_shape=shape, bps=bps,
            complex_data=complex_data, squeeze=squeeze, subset=subset,
            fill_value=0., verify=verify)
        self._payload_nbytes = payload_nbytes
        self._rawdump = rawdump
        self._nfiles = nfiles
        # (the number of frames follows from the last usable timestamp line, looked
        # for when first asked for: `_last_header`, `_nsample`)
        self._lines = lines
        self._start_time = header0.time
        if rawdump:
            self._images = [[host_image(fh_raw)]]
        else:
            self._images = [[host_image(fh) for fh in pair] for pair in fh_raw]
            # phased data are thread-interleaved frames with one slot per
            # polarisation: a channel subset is folded into that decode
            self._plan_channel_select(self.subset, payload_nbytes=payload_nbytes)

    @property
    def payload_nbytes(self):
        return self._payload_nbytes

    def _check_first_frame(self):
        """EOFError unless every raw file holds the first payload whole (what reading
        frame 0 needs: gsb/frame.py fromfile in the reference)."""
        for pair in self._images:
            for img in pair:
                if len(img) < self._payload_nbytes:
                    raise EOFError("could not get full payload of {} bytes: raw file has {}"
                                   .format(self._payload_nbytes, len(img)))

    def _find_last_header(self):
        """Last header of the timestamp file; one that is cut short or does not
        parse is passed over for the one before it, with a warning
        (gsb/base.py:333-372)."""
        import warnings
        lines, h0 = self._lines, self.header0
        last_line = lines[-1]
        try:
            if len(" ".join(last_line.split())) < len(" ".join(h0.words)):
                raise EOFError
            found = GSBHeader(last_line.split())
            found.time
        except Exception:
            warnings.warn("The last header entry, '{0}', has an incorect "
                          "length. Using the second-to-last entry instead.".format(last_line))
            found = GSBHeader(lines[-2].split()) if len(lines) > 1 else h0
        return found

    def _count_samples(self):
        # by the TIMES of the first and the last header for both modes, as the
        # reference counts (base/base.py:827-841): a phased stream opened with half
        # of its raw files then claims twice the samples the files hold, which
        # `info.consistent` reports (gsb/file_info.py:130-160)
        last, h0 = self._last_header, self.header0
        dt = float((last.time - h0.time) / np.timedelta64(1, 'ns')) * 1e-9
        return int(round(dt * self.sample_rate)) + self.samples_per_frame

    @property
    def info(self):
        """Timestamps and raw files together (gsb/file_info.py:98-184): standard
        stream information plus `bandwidth`, `n_raw`, `payload_nbytes` and whether
        the raw files are as long as the timestamps say (`consistent`)."""
        from .info import GSBStreamReaderInfo
        cached = self.__dict__.get('_info')
        if cached is None or cached.closed != self.closed:
            cached = self.__dict__['_info'] = GSBStreamReaderInfo(self)
        return cached

    def close(self):
        self._closed = True
        self._drop_windows()
        if self._pipeline is not None:
            self._pipeline.release()
            self._pipeline = None
        images, self._images, self._set_image = self._images, [], None
        for pair in images:                 # (large mappings are torn down in the background)
            for img in pair:
                retire_image(img)
        del images
        self.fh_ts.close()
        if self._rawdump:
            self.fh_raw.close()
        else:
            for pair in self.fh_raw:
                for fh in pair:
                    fh.close()
        self._unregister()

    # -- staging: the base class streams "frame sets" of a byte image through
    # pinned buffers (staging.WindowPipeline); here set k is block k of every
    # raw file, [pol][part] order, presented as one virtual image
    _file_offset0 = 0

    def _raw_follows(self):
        pass                            # (several raw files and a timestamp file: no one pointer to place)

    def _image(self):
        if getattr(self, '_set_image', None) is None:
            self._set_image = _BlockSetImage(self._images, self._payload_nbytes)
        return self._set_image

    @property
    def _set_nbytes(self):
        return self._payload_nbytes * len(self._images) * self._nfiles

    def _process_window(self, dbuf, first, last, out_flat):
        """Decode sets [first, last) staged in `dbuf` ([set][pol][part][block])."""
        nsets = last - first
        pn = self._payload_nbytes
        npol = len(self._images)
        F = self._nfiles
        nchan = self._unsliced_shape[-1]
        chunk = nchan * (2 if self.complex_data else 1)
        if self._rawdump:
            kernels.decode_frames(dbuf, nsets, pn, _lib.CODER_INT, self.bps,
                                  chunk=chunk, src0=0, src_stride=pn, out=out_flat)
            return
        # output frame (k, part f), slot = polarisation p
        k = torch.arange(nsets, device=dbuf.device, dtype=torch.int64)[:, None, None]
        f = torch.arange(F, device=dbuf.device, dtype=torch.int64)[None, :, None]
        p = torch.arange(npol, device=dbuf.device, dtype=torch.int64)[None, None, :]
        dsrc = (((k * npol + p) * F + f) * pn).reshape(-1).contiguous()
        kernels.decode_frames(dbuf, nsets * F, pn, _lib.CODER_INT, self.bps,
                              chunk=chunk, nslot=npol, src=dsrc,
                              complex_data=self.complex_data, out=out_flat, within=self._within)


class _BlockSetImage:
    """Virtual byte image over the raw files of a GSB observation: set k is
    block k (`pn` bytes) of every file in [pol][part] order.  `pieces(lo, hi)`
    yields the mapped pieces of a byte range (what `staging._stage` copies into
    the pinned buffer); slicing gives a copy."""

    def __init__(self, images, pn):
        self.parts = [im for pair in images for im in pair]
        self.pn = pn
        self.nset = min(len(im) for im in self.parts) // pn
        self.set_nbytes = pn * len(self.parts)

    def __len__(self):
        return self.nset * self.set_nbytes

    def pieces(self, lo, hi):
        pn, m = self.pn, len(self.parts)
        pos = lo
        while pos < hi:
            blk, within = divmod(pos, pn)
            k, j = divmod(blk, m)
            n = min(hi - pos, pn - within)
            yield self.parts[j][k * pn + within:k * pn + within + n]
            pos += n

    def __getitem__(self, item):
        lo, hi, step = item.indices(len(self))
        assert step == 1
        out = np.empty(max(hi - lo, 0), dtype=np.uint8)
        o = 0
        for part in self.pieces(lo, hi):
            out[o:o + len(part)] = part
            o += len(part)
        return out


class GSBStreamWriter(GPUStreamWriterBase):
    """GSB stream writer (gsb/base.py:388-457): one timestamp line per frame
    to `fh_ts`; samples are rounded / clipped / packed on the GPU (signed
    4-bit nibbles for rawdump, int8 pairs for phased) and written to one raw
    file (rawdump) or to ``fh_raw[pol][part]`` (phased: parts are consecutive
    in time).  Arguments and defaults as for the reader."""
    _sample_shape_fields = staticmethod(lambda n: ('nchan',) if n == 1 else ('nthread', 'nchan'))
    __repr__ = _stream_repr

    def __init__(self, fh_ts, fh_raw, header0=None, sample_rate=None,
                 samples_per_frame=None, payload_nbytes=None, nchan=None, bps=None,
                 complex_data=None, squeeze=True, mode=None, **kwargs):
        if header0 is None:
            header0 = GSBHeader.fromvalues(mode, **kwargs)
        elif kwargs:
            raise TypeError("got unexpected arguments {}".format(sorted(kwargs)))
        self.fh_ts = fh_ts
        sample_rate = hz(sample_rate)
        rawdump = header0.mode == 'rawdump'
        fh_raw = _raw_handles(fh_raw, rawdump)
        complex_data = (not rawdump) if complex_data is None else complex_data
        bps = bps if bps is not None else (4 if rawdump else 8)
        nchan = nchan if nchan is not None else (1 if rawdump else 512)
        bpfs = bps * nchan * (2 if complex_data else 1)
        nfiles = 1 if rawdump else len(fh_raw[0])
        if payload_nbytes is None:
            if samples_per_frame is not None:
                payload_nbytes = samples_per_frame * bpfs // (8 * nfiles)
            elif sample_rate is None:
                payload_nbytes = 2 ** 22
            else:
                payload_nbytes = int(round(sample_rate / DEFAULT_FRAME_RATE * bpfs / 8 / nfiles))
        if samples_per_frame is None:
            samples_per_frame = payload_nbytes * 8 // bpfs * nfiles
        elif samples_per_frame != payload_nbytes * nfiles * 8 / bpfs:
            raise ValueError('inconsistent samples_per_frame, bps, '
                             'complex_data, and payload_nbytes')
        if sample_rate is None:
            sample_rate = samples_per_frame * DEFAULT_FRAME_RATE
        shape = (nchan,) if rawdump else (len(fh_raw), nchan)
        super().__init__(fh_raw, header0, sample_rate=hz(sample_rate),
                         samples_per_frame=samples_per_frame, unsliced_shape=shape,
                         bps=bps, complex_data=complex_data, squeeze=squeeze)
        self._payload_nbytes, self._rawdump, self._nfiles = payload_nbytes, rawdump, nfiles
        self._start_time = header0.time

    payload_nbytes = property(lambda self: self._payload_nbytes)

    def _frame_header(self, index):
        """Timestamp of frame `index` (gsb/base.py:220-229): times advance by
        the frame duration, the sequence number by one, the memory block
        cycles modulo 8."""
        h0 = self.header0
        step = np.timedelta64(int(round(index * self.samples_per_frame * 1e9 / self.sample_rate)), 'ns')
        if self._rawdump:
            return GSBHeader.fromvalues('rawdump', time=h0.time + step)
        return GSBHeader.fromvalues('phased', gps_time=h0.gps_time + step,
                                    pc_time=h0.pc_time + step,
                                    seq_nr=h0['seq_nr'] + index,
                                    mem_block=(h0['mem_block'] + index) % 8)

    def _write_frames(self, data, valid):
        spf = self.samples_per_frame
        nframes = data.shape[0] // spf
        if data.is_complex():
            data = torch.view_as_real(data)
        if self._rawdump:
            write_device_bytes(self.fh_raw, kernels.encode_flat(data, _lib.CODER_INT, self.bps))
        else:
            npol, F = len(self.fh_raw), self._nfiles
            # (frame, part, time in part, pol, ...) -> (pol, part, frame, time, ...)
            block = data.reshape((nframes, F, spf // F, npol) + tuple(data.shape[2:]))
            block = block.permute(3, 1, 0, 2, *range(4, block.dim()))
            packed = kernels.encode_flat(block, _lib.CODER_INT, self.bps)
            packed = packed.reshape(npol, F, nframes * self._payload_nbytes)
            for p in range(npol):
                for f in range(F):
                    write_device_bytes(self.fh_raw[p][f], packed[p, f])
        for k in range(nframes):
            self._frame_header(self._nframes_written + k).tofile(self.fh_ts)

    def flush(self):
        self.fh_ts.flush()
        super().flush()                 # (waits for the queued pieces of every raw file, then flushes it)

    def _close_files(self):
        self.fh_ts.close()
        for fh in ([self.fh_raw] if self._rawdump else [f for pair in self.fh_raw for f in pair]):
            fh.close()


def _name_of(fh):
    """File name(s) of a handle or a nest of handles; None if any has none."""
    if isinstance(fh, (tuple, list)):
        names = tuple(_name_of(f) for f in fh)
        return None if any(n is None for n in names) else names
    name = getattr(fh, 'name', None)
    return name if isinstance(name, str) else None


def _reopen_stream(opener, source, kwargs, offset, closed=False):
    """Unpickling / copying a stream reader: open the files again, go to `offset`;
    a reader that was closed comes back closed."""
    reader = open(source[0], 'rs', raw=source[1], **kwargs)
    reader.offset = offset
    if closed:
        reader.close()
    return reader


def open(name, mode='rs', **kwargs):
    """Open a GSB timestamp file plus ``raw=`` data file(s) for stream reading
    (``'rs'``) or writing (``'ws'``) (gsb/base.py:460-560).  ``raw`` is one
    file for rawdump, a (nested) tuple ``((polL1, polL2), (polR1, polR2))``
    for phased data; ``header_mode`` overrides the mode inferred from it."""
    from ..base.quantities import normalize_kwargs
    kwargs = normalize_kwargs(kwargs)      # (Quantity rates, Time instants of reference callers)
    if mode in ('rt', 'wt'):
        fh = name if hasattr(name, 'read') or hasattr(name, 'write') else io.open(name, mode[0])
        return GSBTimeStampIO(fh, **kwargs)
    if mode == 'rb':
        return GSBFileReader(name if hasattr(name, 'read') else io.open(name, 'rb'), **kwargs)
    if mode == 'wb':
        return GSBFileWriter(name if hasattr(name, 'write') else LazyWriteFile(name), **kwargs)
    if mode not in ('rs', 'ws'):
        raise ValueError("supported modes are 'rt', 'wt', 'rb', 'wb', 'rs' and 'ws' "
                         "(got {!r}).".format(mode))
    raw = kwargs.pop('raw', None)
    if raw is None:
        raise TypeError("stream missing required argument 'raw'.")
    stream_mode = kwargs.pop('header_mode',
                             'phased' if isinstance(raw, (tuple, list)) else 'rawdump')
    rw = mode[0]
    attr = 'read' if rw == 'r' else 'write'

    def handle(f, text=False):
        if hasattr(f, attr):
            return f
        return io.open(f, rw + ('' if text else 'b')) if rw == 'r' else (
            io.open(f, 'w') if text else LazyWriteFile(f))

    fh_ts = handle(name, text=True)
    if stream_mode == 'rawdump':
        fh_raw = handle(raw)
    else:
        if not isinstance(raw, (tuple, list)):
            raw = ((raw,),)
        elif not isinstance(raw[0], (tuple, list)):
            raw = (raw,)
        fh_raw = tuple(tuple(handle(f) for f in pair) for pair in raw)
    if rw == 'r':
        reader = GSBStreamReader(fh_ts, fh_raw, **kwargs)
        # pickling / copying: the files are opened again by name on arrival
        source = (_name_of(fh_ts), _name_of(fh_raw))
        if source[0] is not None and source[1] is not None:
            reader._pickle_recipe = (_reopen_stream, None, source,
                                     dict(kwargs, header_mode=stream_mode))
        return reader
    if 'header0' not in kwargs:
        kwargs['mode'] = stream_mode
    return GSBStreamWriter(fh_ts, fh_raw, **kwargs)
