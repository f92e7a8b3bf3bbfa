"""GUPPI file and stream readers, and ``open`` (guppi/base.py:27-278)."""


import numpy as np

from .. import _lib, kernels
from ..base.base import FileBase, VLBIFileReaderBase
from ..base.blockreader import BlockStreamReader
from ..base.opener import FormatOpener
from ..base.writer import BlockStreamWriter
from ..helpers.sequentialfile import UpperCaseSequencer
from .header import GUPPIHeader
from .payload import GUPPIPayload
from .frame import GUPPIFrame

__all__ = ['GUPPIFileNameSequencer', 'GUPPIFileReader', 'GUPPIStreamReader',
           'GUPPIStreamWriter', 'open', 'GUPPIFileWriter']

# template fields are matched to the (upper-case) header keys ignoring case
# (guppi/base.py:23-85)
GUPPIFileNameSequencer = UpperCaseSequencer


class GUPPIFileReader(VLBIFileReaderBase):
    _format = 'guppi'

    def _info_extras(self, header0, offset0):
        extras = {'pktfmt': header0['PKTFMT'], 'overlap': header0.overlap}
        if extras['pktfmt'] not in header0.supported_formats:
            # (read as the usual layout all the same: guppi/file_info.py:28-35 in the reference)
            extras['_warnings'] = {'pktfmt': 'Unknown pktfmt {!r}. Assuming channels are stored first.'
                                   .format(extras['pktfmt'])}
        return extras

    def read_header(self):
        return GUPPIHeader.fromfile(self.fh_raw)

    def read_frame(self, memmap=True, verify=True):
        return GUPPIFrame.fromfile(self.fh_raw, memmap=memmap, verify=verify)

    def get_frame_rate(self):
        with self.temporary_offset(0):
            header = self.read_header()
        return header.sample_rate / (header.samples_per_frame - header.overlap)


class GUPPIFileWriter(FileBase):
    """Frame-level writer (guppi/base.py): header + payload packed on the GPU."""

    def write_frame(self, data, header=None, **kwargs):
        if not isinstance(data, GUPPIFrame):
            if header is None:
                header = GUPPIHeader.fromvalues(**kwargs)
            data = GUPPIFrame.fromdata(data, header)
        return data.tofile(self.fh_raw)

    def memmap_frame(self, header=None, **kwargs):
        """Write the header now and map the payload, so that the frame can be
        filled in pieces by setting slices of it (guppi/base.py `memmap_frame`);
        every piece is packed by the GPU encoder."""
        if header is None:
            header = GUPPIHeader.fromvalues(**kwargs)
        header.tofile(self.fh_raw)
        payload = GUPPIPayload.fromfile(self.fh_raw, memmap=True, header=header)
        return GUPPIFrame(header, payload)


class GUPPIStreamReader(BlockStreamReader):
    """GUPPI stream -> device tensor (nsample, npol, nchan)."""
    _sample_shape_fields = ('npol', 'nchan')

    def __init__(self, fh_raw, squeeze=True, subset=(), verify=True):
        fh_raw = GUPPIFileReader(fh_raw)
        header0 = fh_raw.read_header()
        super().__init__(
            fh_raw, header0, sample_rate=header0.sample_rate,
            samples_per_frame=header0.samples_per_frame - header0.overlap,
            unsliced_shape=header0.sample_shape, bps=header0.bps,
            complex_data=header0.complex_data, squeeze=squeeze, subset=subset,
            fill_value=0., verify=verify)
        self._header_nbytes = header0.nbytes
        self._frame_nbytes = header0.frame_nbytes
        self._file_offset0 = 0
        self._nframes = len(self._image()) // self._frame_nbytes
        self._spf_full = header0.samples_per_frame
        self._overlap = header0.overlap
        self._nsample = self._nframes * self.samples_per_frame + self._overlap
        self._start_time = header0.time
        if self.complex_data and self.bps == 8:
            self._plan_channel_range()

    def _image(self):
        return self.fh_raw.image()

    @property
    def _frame_rate(self):
        return self.sample_rate / self.samples_per_frame        # (frames advance by samples_per_frame - overlap)

    def _find_last_header(self):
        with self.fh_raw.temporary_offset((self._nframes - 1) * self._frame_nbytes):
            return self.fh_raw.read_header()

    def _pieces(self, offset, count):
        """The reference loop: the frame a read starts in is taken up to its
        END (overlap tail included); following frames are entered at their
        sample OVERLAP (base/base.py:957-967; guppi/base.py:270-278)."""
        keep, spf = self.samples_per_frame, self._spf_full
        normal_end = self._nsample - self._overlap
        pieces, done = [], 0
        while done < count:
            o = offset + done
            if normal_end <= o < self._nsample:
                index, so = divmod(normal_end - 1, keep)
                so += 1 + o - normal_end
            else:
                index, so = divmod(o, keep)
            n = min(count - done, spf - so)
            pieces.append((index, so, so + n))
            done += n
        return pieces

    def _row_range_source(self, frame, a, b):
        """Times [a, b) of a block: one run per channel (channels first) or one
        run in all (time first); staged back to back they are a block of
        b - a times in the same storage order."""
        h = self.header0
        if self.bps != 8 or not self.complex_data:
            return None
        if self._sel is not None:
            return None                 # (a channel list / one polarisation: whole blocks, `_tiled_decode`)
        npol, nchan, T = h.npol, h.nchan, self._spf_full
        base = self._frame_span(frame)[0] + self._header_nbytes
        step = npol * 2                                   # bytes per time of one channel
        c0, nkeep = self._chan_lo, self._decode_shape[-1]           # (the channels asked for)
        if h.channels_first:
            if nkeep > 4096:
                return None
            # only the kept channels' runs are staged: together they are a
            # block of `nkeep` channels
            pieces = [(base + (c * T + a) * step, (b - a) * step) for c in range(c0, c0 + nkeep)]
            layout, stored, skip = _lib.LAYOUT_GUPPI_CF, 0, 0
        else:
            # whole times are staged; the kernel enters each at channel c0
            pieces = [(base + a * nchan * step, (b - a) * nchan * step)]
            layout, stored, skip = _lib.LAYOUT_GUPPI_TF, nchan, c0 * step

        def decode(dbuf, out_flat):
            kernels.decode_i8_tiled(dbuf, 1, layout, npol, nkeep, b - a, 0, b - a,
                                    src0=skip, out=out_flat, nchan_stored=stored)
        return pieces, decode

    def _decode_window(self, dbuf, nframes, a, b, out_flat, payload_offset,
                       frame_stride, first_frame):
        h = self.header0
        if self.bps != 8:
            raise KeyError(self.bps)
        if not self.complex_data:
            npol = h.npol
            for i in range(nframes):
                src = payload_offset + i * frame_stride
                b0, b1 = a * npol, b * npol
                lo, hi = b0 - b0 % 4, -(-b1 // 4) * 4
                tmp = kernels.decode_frames(dbuf, 1, hi - lo, _lib.CODER_INT, 8,
                                            src0=src + lo)
                n = (b - a) * npol
                out_flat[i * n:(i + 1) * n] = tmp[b0 - lo:b1 - lo]
            return
        layout = _lib.LAYOUT_GUPPI_CF if h.channels_first else _lib.LAYOUT_GUPPI_TF
        # (a planned channel range or selection is applied by the decode)
        self._tiled_decode(dbuf, nframes, layout, self._spf_full, a, b, payload_offset, frame_stride, out_flat)


class GUPPIStreamWriter(BlockStreamWriter):
    """GUPPI stream writer (guppi/base.py:281-310): no overlap; frame k gets
    ``PKTIDX = header0's + k * packets per frame`` (guppi/base.py:209-225)."""
    _sample_shape_fields = ('npol', 'nchan')

    def __init__(self, fh_raw, header0=None, squeeze=True, **kwargs):
        if header0 is None:
            header0 = GUPPIHeader.fromvalues(**kwargs)
        elif kwargs:
            raise TypeError("got unexpected arguments {}".format(sorted(kwargs)))
        assert header0.get('OVERLAP', 0) == 0, "overlap must be 0 when writing GUPPI files."
        if header0.bps != 8:
            raise ValueError("GUPPIPayload cannot encode data with {} bits".format(header0.bps))
        super().__init__(fh_raw, header0, sample_rate=header0.sample_rate,
                         samples_per_frame=header0.samples_per_frame,
                         unsliced_shape=header0.sample_shape, bps=header0.bps,
                         complex_data=header0.complex_data, squeeze=squeeze)
        self._start_time = header0.time
        self._packets_per_frame = header0.payload_nbytes // header0['PKTSIZE']

    def _frame_header(self, index):
        header = self.header0.copy()
        header['PKTIDX'] = self.header0['PKTIDX'] + index * self._packets_per_frame
        return header

    def _storage_order(self, block):
        # (frame, time, pol, chan[, re/im]) -> (frame, chan, time, pol[, re/im]) or
        # (frame, time, chan, pol[, re/im]) (guppi/payload.py:90-102 inverted)
        if self.header0.channels_first:
            return block.movedim(3, 1)
        return block.transpose(2, 3)


class _GUPPIOpener(FormatOpener):
    def __call__(self, name, mode='rs', **kwargs):
        per_file = kwargs.pop('frames_per_file', 128)
        from ..base.opener import source_kind
        if self.normalize_mode(mode) == 'ws' and source_kind(name) in ('sequence', 'template'):
            if kwargs.get('header0') is None:
                # header keywords instead of a header: make it here, so that the files of
                # the sequence get their size from it (guppi/base.py:351-368 in the reference)
                squeeze = kwargs.pop('squeeze', True)
                extra = {k: kwargs.pop(k) for k in ('file_size',) if k in kwargs}
                kwargs = dict(header0=GUPPIHeader.fromvalues(**kwargs), squeeze=squeeze, **extra)
            if 'file_size' not in kwargs:
                kwargs['file_size'] = per_file * kwargs['header0'].frame_nbytes
        return super().__call__(name, mode, **kwargs)


def _adopt_header(h):
    """The reference's GUPPIHeader (an `astropy.io.fits.Header`) -> ours: the
    cards' keywords, values and comments."""
    if isinstance(h, GUPPIHeader) or not hasattr(h, 'cards'):
        return h
    cards = [(c.keyword, c.value, c.comment) for c in h.cards if c.keyword not in ('', 'COMMENT', 'HISTORY', 'END')]
    new = GUPPIHeader({k: (bool(v) if isinstance(v, (bool, np.bool_)) else v) for k, v, _ in cards}, verify=False,
                      mutable=True)
    new.comments = {k: c for k, _, c in cards if c}
    return new


open = _GUPPIOpener('GUPPI', {'rb': GUPPIFileReader, 'wb': GUPPIFileWriter,
                            'rs': GUPPIStreamReader,
                              'ws': GUPPIStreamWriter},
                    sequencer=GUPPIFileNameSequencer, adopt_header=_adopt_header)
open.__doc__ = """Open GUPPI raw file(s) (guppi/base.py:305-396): ``'rb'``, ``'rs'`` or
``'ws'``; names, handles, lists of names, or a template such as
``'puppi_{stt_imjd}_{src_name}_{scannum}.{file_nr:04d}.raw'``.  A written
sequence gets ``frames_per_file`` (default 128) frames per file."""
