"""DADA file and stream readers, and ``open`` (dada/base.py:99-330)."""
import io
from math import gcd

import numpy as np
import torch

from .. import _lib, kernels
from ..base.base import FileBase, VLBIFileReaderBase
from ..base.blockreader import BlockStreamReader
from ..base.opener import FormatOpener, source_kind
from ..base.writer import BlockStreamWriter
from ..helpers.sequentialfile import UpperCaseSequencer
from .header import DADAHeader
from .payload import DADAPayload, decode_i8_rows
from .frame import DADAFrame

__all__ = ['DADAFileNameSequencer', 'DADAFileReader', 'DADAStreamReader',
           'DADAStreamWriter', 'open', 'DADAFileWriter']


class DADAFileNameSequencer(UpperCaseSequencer):
    """DADA file names from a template (dada/base.py:27-96): fields match the
    header keys ignoring case, and ``{obs_offset}`` advances by the header's
    ``FILE_SIZE`` for every file.

    >>> DADAFileNameSequencer('{date}_{file_nr:03d}.dada', {'DATE': "2018-01-01"})[10]
    '2018-01-01_010.dada'
    """

    def __init__(self, template, header={}):
        super().__init__(template, header)
        self._step = header['FILE_SIZE'] if 'OBS_OFFSET' in self.items else 0

    def _values(self, file_nr):
        values = super()._values(file_nr)
        if self._step:
            values['OBS_OFFSET'] = self.items['OBS_OFFSET'] + file_nr * self._step
        return values


class DADAFileReader(VLBIFileReaderBase):
    _format = 'dada'

    def read_header(self):
        return DADAHeader.fromfile(self.fh_raw)

    def read_frame(self, memmap=True, verify=True):
        return DADAFrame.fromfile(self.fh_raw, memmap=memmap, verify=verify)

    def get_frame_rate(self):
        with self.temporary_offset(0):
            header = self.read_header()
        return header.sample_rate / header.samples_per_frame


class DADAFileWriter(FileBase):
    """Frame-level writer (dada/base.py): header + payload packed on the GPU."""

    def write_frame(self, data, header=None, **kwargs):
        if not isinstance(data, DADAFrame):
            if header is None:
                header = DADAHeader.fromvalues(**kwargs)
            data = DADAFrame.fromdata(data, header)
        return data.tofile(self.fh_raw)

    def memmap_frame(self, header=None, **kwargs):
        """Write the header now and map the payload, so that the frame can be
        filled in pieces by setting slices of it (dada/base.py `memmap_frame`);
        every piece is packed by the GPU encoder."""
        if header is None:
            header = DADAHeader.fromvalues(**kwargs)
        header.tofile(self.fh_raw)
        payload = DADAPayload.fromfile(self.fh_raw, memmap=True, header=header)
        return DADAFrame(header, payload)


class DADAStreamReader(BlockStreamReader):
    """DADA stream -> device tensor (nsample, npol, nchan)."""
    _sample_shape_fields = ('npol', 'nchan')

    def __init__(self, fh_raw, squeeze=True, subset=(), verify=True):
        fh_raw = DADAFileReader(fh_raw)
        header0 = fh_raw.read_header()
        super().__init__(
            fh_raw, header0, sample_rate=header0.sample_rate,
            samples_per_frame=header0.samples_per_frame,
            unsliced_shape=header0.sample_shape, bps=header0.bps,
            complex_data=header0.complex_data, squeeze=squeeze, subset=subset,
            fill_value=0., verify=verify)
        self._header_nbytes = header0.nbytes
        self._frame_nbytes = header0.frame_nbytes
        self._file_offset0 = 0
        self._mkbf = header0.get('INSTRUMENT') == 'MKBF'
        size = len(self._image())
        # a truncated last frame counts down to whole payload blocks
        # (dada/base.py:251-306)
        nframes, partial = divmod(size, self._frame_nbytes)
        self._row_nbytes = header0._sample_nbits // 8
        self._last_rows = self.samples_per_frame
        if partial > self._header_nbytes:
            nframes += 1
            block = 4 * self._row_nbytes // gcd(4, self._row_nbytes)
            self._last_rows = ((partial - self._header_nbytes) // block * block
                               // self._row_nbytes)
            if nframes == 1:
                self.samples_per_frame = self._last_rows
        elif nframes == 0:
            raise EOFError('file (of {0} bytes) appears to end without'
                           'any payload.'.format(partial))
        self._nframes = nframes
        self._nsample = (nframes - 1) * header0.samples_per_frame + self._last_rows
        self._spf0 = header0.samples_per_frame
        self._start_time = header0.time
        if nframes == 1 and self._last_rows != header0.samples_per_frame:
            # one frame, cut short: the stream's header is that frame's header with the
            # payload size the file really holds (dada/base.py:266-270)
            self._header0 = self._last_header
        if self.bps == 8 and not self._mkbf:
            # plain DADA samples are (pol, chan) runs of int8: a subset that
            # keeps every polarisation's same channels is folded into the decode
            self._plan_channel_select(self.subset, lead_in_sample=True)
        elif self.bps == 8:
            # MKBF heaps: a channel range is the same transpose over fewer channels
            self._plan_channel_range()

    def _image(self):
        return self.fh_raw.image()

    @property
    def _frame_rate(self):
        return self.sample_rate / self._spf0

    def _find_last_header(self):
        """Header of the last frame; when that frame is cut short, a mutable copy whose
        payload size is what the file holds in whole words and complete samples
        (dada/base.py:277-306)."""
        with self.fh_raw.temporary_offset((self._nframes - 1) * self._frame_nbytes):
            header = self.fh_raw.read_header()
        if self._last_rows != self._spf0:
            header = header.copy()
            header.payload_nbytes = self._last_rows * self._row_nbytes
        return header

    def _frame_span(self, frame):
        lo = frame * self._frame_nbytes
        return lo, min(self._frame_nbytes, len(self._image()) - lo)

    def _pieces(self, offset, count):
        spf = self._spf0
        pieces, done = [], 0
        while done < count:
            index, so = divmod(offset + done, spf)
            rows = self._last_rows if index == self._nframes - 1 else spf
            n = min(count - done, rows - so)
            pieces.append((index, so, so + n))
            done += n
        return pieces

    def _row_range_source(self, frame, a, b):
        """Rows [a, b) of a frame are contiguous (whole heaps for MKBF)."""
        if self.bps != 8:
            return None
        base = self._frame_span(frame)[0] + self._header_nbytes
        npol, nchan = self._unsliced_shape
        if self._mkbf and self._sel is not None:
            return None                 # (a channel list / one polarisation: whole frames, `_tiled_decode`)
        if self._mkbf:
            h0, h1 = a // 256, -(-b // 256)
            heap = npol * nchan * 256 * 2
            pieces = [(base + h0 * heap, (h1 - h0) * heap)]
            nkeep = self._decode_shape[-1]
            skip = kernels.tiled_channel_skip(_lib.LAYOUT_MKBF, npol, 0, self._chan_lo)

            def decode(dbuf, out_flat):
                kernels.decode_i8_tiled(dbuf, 1, _lib.LAYOUT_MKBF, npol, nkeep, (h1 - h0) * 256,
                                        a - h0 * 256, b - h0 * 256, src0=skip, out=out_flat,
                                        nchan_stored=nchan)
            return pieces, decode
        rb = self._row_nbytes
        pieces = [(base + a * rb, (b - a) * rb)]

        def decode(dbuf, out_flat):
            self._flat_rows(dbuf, 1, 0, (b - a) * rb, 0, out_flat)
        return pieces, decode

    def _flat_rows(self, dbuf, nframes, lo, n, stride, out_flat):
        """`n` bytes of whole samples from each of `nframes` frames, at ``lo +
        i * stride`` -> float32 values in `out_flat`; with a planned channel
        selection only the kept positions of every sample are written."""
        within = self._within
        rb = self._row_nbytes                   # int8 values per sample
        if n % 4 == 0 and lo % 4 == 0 and stride % 4 == 0:
            # rows of every frame in ONE launch: `nframes` payloads of n bytes
            # at a fixed stride
            try:
                kernels.decode_frames(dbuf, nframes, n, _lib.CODER_INT, 8,
                                      chunk=rb if within is not None else 1, src0=lo,
                                      src_stride=stride, out=out_flat, within=within)
                return
            except (KeyError, _lib.BBError):
                if within is None:
                    raise                       # (a short request the selecting kernel cannot
                    # cut into whole rows: decode the rows, index below)
        per = n if within is None else n // rb * within.numel()
        for i in range(nframes):
            rows = decode_i8_rows(dbuf, lo + i * stride, rb, 0, n // rb)
            if within is not None:
                rows = rows.view(-1, rb)[:, within.long()].reshape(-1)
            out_flat[i * per:(i + 1) * per] = rows

    def _decode_window(self, dbuf, nframes, a, b, out_flat, payload_offset,
                       frame_stride, first_frame):
        if self.bps not in (8, 32) or (self._mkbf and self.bps != 8):
            raise KeyError(self.bps)
        npol, nchan = self._unsliced_shape
        if self._mkbf:
            # a truncated last frame only holds `_last_rows` samples (whole heaps)
            n_full = nframes
            row = (b - a) * int(np.prod(self._decode_shape)) * 2       # (a planned range / selection: fewer values)
            if first_frame + nframes == self._nframes and self._last_rows < self._spf0:
                n_full -= 1
                self._tiled_decode(dbuf, 1, _lib.LAYOUT_MKBF, self._last_rows // 256 * 256, a, b,
                                   payload_offset + n_full * frame_stride, 0, out_flat[n_full * row:])
            if n_full:
                self._tiled_decode(dbuf, n_full, _lib.LAYOUT_MKBF, self._spf0, a, b, payload_offset,
                                   frame_stride, out_flat[:n_full * row])
            return
        if self.bps == 32:
            # EXTENSION (no counterpart in the reference, which knows NBIT 8
            # only: dada/payload.py:40-41): float32 samples are passed through
            # as they are -- ONE strided copy launch of the library
            # (bb_copy_frames, k_copy.h), byte-identical to the file
            nb = (b - a) * self._row_nbytes
            lo = payload_offset + a * self._row_nbytes
            if lo % 4 == 0 and frame_stride % 4 == 0 and dbuf.data_ptr() % 4 == 0:
                kernels.copy_frames(dbuf, nframes, nb, src0=lo, src_stride=frame_stride, out=out_flat)
                return
            for i in range(nframes):                    # (an odd header size: byte moves by torch)
                out_flat[i * nb // 4:(i + 1) * nb // 4] = \
                    dbuf[lo + i * frame_stride:lo + i * frame_stride + nb].clone().view(torch.float32)
            return
        self._flat_rows(dbuf, nframes, payload_offset + a * self._row_nbytes,
                        (b - a) * self._row_nbytes, frame_stride, out_flat)


class DADAStreamWriter(BlockStreamWriter):
    """DADA stream writer (dada/base.py:333-362): ``header0`` describes the
    first frame; frame k gets ``OBS_OFFSET = header0's + k * payload_nbytes``
    (dada/base.py:222-225).  Without ``header0`` the keywords make one."""
    _sample_shape_fields = ('npol', 'nchan')

    def __init__(self, fh_raw, header0=None, squeeze=True, **kwargs):
        if header0 is None:
            header0 = DADAHeader.fromvalues(**kwargs)
        elif kwargs:
            raise TypeError("got unexpected arguments {}".format(sorted(kwargs)))
        assert header0.get('OBS_OVERLAP', 0) == 0
        if header0.bps != 8:
            raise ValueError("DADAPayload cannot encode data with {} bits".format(header0.bps))
        super().__init__(fh_raw, header0, sample_rate=header0.sample_rate,
                         samples_per_frame=header0.samples_per_frame,
                         unsliced_shape=header0.sample_shape, bps=header0.bps,
                         complex_data=header0.complex_data, squeeze=squeeze)
        self._start_time = header0.time

    def _frame_header(self, index):
        header = self.header0.copy()
        header['OBS_OFFSET'] = self.header0['OBS_OFFSET'] + index * self.header0.payload_nbytes
        return header

    def _storage_order(self, block):
        if self.header0.get('INSTRUMENT') != 'MKBF':
            return block
        # MeerKAT beamformer: heaps of 256 samples, stored (heap, pol, chan, 256, re/im)
        # (dada/payload.py:54-89 in the reference; the reader's `bb_decode_i8_tiled` undoes it)
        nframes, spf, npol, nchan = block.shape[:4]
        return block.reshape(nframes, spf // 256, 256, npol, nchan, -1).movedim(2, 4)


class _DADAOpener(FormatOpener):
    def __call__(self, name, mode='rs', **kwargs):
        if (self.normalize_mode(mode) == 'ws' and kwargs.get('header0') is None
                and source_kind(name) in ('sequence', 'template')):
            # header keywords instead of a header: make it here, so that the files of the
            # sequence get their size (one frame each) from it (dada/base.py:411-425 there)
            squeeze = kwargs.pop('squeeze', True)
            extra = {k: kwargs.pop(k) for k in ('file_size',) if k in kwargs}
            kwargs = dict(header0=DADAHeader.fromvalues(**kwargs), squeeze=squeeze, **extra)
        return super().__call__(name, mode, **kwargs)

    def _sequencer_for(self, template, mode, kwargs):
        extra = {}
        if mode[0] == 'r' and 'obs_offset' in template.lower() and kwargs.get('header0') is None:
            # (the reference fills the template from a header made of the keywords, whose defaults
            # put the first file at offset 0: base/base.py:1765-1779)
            given = {k.lower() for k in kwargs}
            extra = {k: 0 for k in ('OBS_OFFSET', 'FILE_SIZE') if k.lower() not in given}
            kwargs.update(extra)
        try:
            fns = super()._sequencer_for(template, mode, kwargs)
        finally:
            for k in extra:
                kwargs.pop(k, None)
        if mode[0] == 'r' and 'obs_offset' in template.lower():
            # the step of {obs_offset} is the size found in the first file
            # (dada/base.py:366-377)
            with io.open(fns[0], 'rb') as fh:
                fns = self.sequencer(template, DADAHeader.fromfile(fh))
        return fns


def _adopt_header(h):
    """The reference's DADAHeader (an ordered mapping with `comments`) -> ours."""
    if isinstance(h, DADAHeader) or not hasattr(h, 'items'):
        return h
    new = DADAHeader({k: v for k, v in h.items() if not str(k).startswith('_')}, verify=False, mutable=True)
    new.comments = {k: v for k, v in dict(getattr(h, 'comments', {}) or {}).items() if not str(k).startswith('_')}
    return new


open = _DADAOpener('DADA', {'rb': DADAFileReader, 'wb': DADAFileWriter,
                            'rs': DADAStreamReader,
                            'ws': DADAStreamWriter},
                   sequencer=DADAFileNameSequencer,
                   default_file_size=lambda header0: header0.frame_nbytes, adopt_header=_adopt_header)
open.__doc__ = """Open DADA file(s) (dada/base.py:366-470): ``'rb'``, ``'rs'`` or ``'ws'``;
names, handles, lists of names, or a template such as
``'{utc_start}_{obs_offset:016d}.000000.dada'``.  A written sequence gets one
frame per file unless ``file_size`` says otherwise."""
