"""Mark 4 file and stream readers, and ``open``.

Mirrors ``Mark4FileReader`` (mark4/base.py:28-207) and ``Mark4StreamReader``
(mark4/base.py:233-312).  Per staged window the stream reader launches
``bb_mark4_scan`` (sync, error flags, BCD time -> frame index) ->
``bb_build_index`` -> ``bb_decode_mark4`` (track demultiplexing + header
fill).
"""
import operator

import numpy as np
import torch

from .. import _lib, kernels
from ..base.base import (FileBase, VLBIFileReaderBase, GPUStreamReaderBase,
                         HeaderNotFoundError)
from .header import Mark4Header, MARK4_DTYPES, frame_header_streams
from .frame import Mark4Frame
from ._bitmaps import BITMAPS
from ..base.writer import GPUStreamWriterBase
from ..base.opener import FormatOpener
from ..base.quantities import hz

__all__ = ['Mark4FileReader', 'Mark4StreamReader', 'Mark4StreamWriter', 'open', 'Mark4FileWriter']


class Mark4FileReader(VLBIFileReaderBase):
    _format = 'mark4'
    _info_find_kwargs = {}                  # the first frame may start anywhere

    def _info_needs(self):
        if self.decade is None and self.ref_time is None:
            return {'decade': "needed to infer full times.",
                    'ref_time': "needed to infer full times."}
        return {}

    def _info_extras(self, header0, offset0):
        return {'ntrack': header0.ntrack, 'offset0': offset0}

    def _info_format_by_search(self):
        # (frames are there although no header could be made of them: a decade that
        # cannot be; mark4/file_info.py decides the format by the frames alone)
        if self.ntrack is None:
            self.determine_ntrack()
        return self.locate_frames()

    def _info_number_of_frames(self, header0, offset0):
        with self.temporary_offset(-header0.frame_nbytes, 2):
            self.find_header(forward=False)
            return (self.fh_raw.tell() - offset0) / header0.frame_nbytes + 1

    def __init__(self, fh_raw, ntrack=None, decade=None, ref_time=None):
        self.ntrack = operator.index(ntrack) if ntrack is not None else None
        self.decade = operator.index(decade) if decade is not None else None
        self.ref_time = ref_time
        super().__init__(fh_raw)

    def __repr__(self):
        return ("{name}(fh_raw={s.fh_raw}, ntrack={s.ntrack}, decade={s.decade}, ref_time={s.ref_time})"
                .format(name=type(self).__name__, s=self))

    def read_header(self):
        return Mark4Header.fromfile(self.fh_raw, ntrack=self.ntrack,
                                    decade=self.decade, ref_time=self.ref_time)

    def read_frame(self, verify=True):
        frame = Mark4Frame.fromfile(self.fh_raw, self.ntrack, decade=self.decade,
                                    ref_time=self.ref_time, verify=verify)
        return self._lend_device_words(frame)

    def _sync_at(self, image, o, ntrack):
        """Sync pattern at frame offset o: stream word 63 zero, words 64-95
        all ones (mark4/header.py:345-373)."""
        dt = np.dtype(MARK4_DTYPES[ntrack])
        isz = dt.itemsize
        if o < 0 or o + 96 * isz > len(image):
            return False
        w = np.frombuffer(image[o + 63 * isz:o + 96 * isz].tobytes(), dtype=dt)
        return bool(w[0] == 0 and np.all(w[1:] == np.iinfo(dt).max))

    def locate_frames(self, pattern=None, *, mask=None, frame_nbytes=None, offset=0,
                      forward=True, maximum=None, check=1, _here_first=False):
        """As `VLBIFileReaderBase.locate_frames`; by default the Mark 4 sync
        pattern of all tracks plus the zero bit before it, for the reader's
        `ntrack` (found with `determine_ntrack` if not known): stream word 63
        zero, words 64-95 all ones (mark4/base.py:110-166)."""
        # a frame is 20000 stream words of ntrack bits = 2500 bytes per track
        if frame_nbytes is not None:
            if frame_nbytes % 2500:
                raise ValueError('Mark 4 frames hold 2500 bytes per track: frame_nbytes '
                                 'must be a multiple of 2500 bytes.')
            ntrack = frame_nbytes // 2500
        else:
            ntrack = self.ntrack
            if ntrack is None:
                with self.temporary_offset(0):
                    ntrack = self.determine_ntrack(maximum=maximum)
            frame_nbytes = 2500 * ntrack
        if pattern is None:
            isz = ntrack // 8
            pattern = np.concatenate([np.zeros(isz, np.uint8), np.full(32 * isz, 0xff, np.uint8)])
            offset = offset + 63 * isz
        return super().locate_frames(pattern, mask=mask, frame_nbytes=frame_nbytes,
                                     offset=offset, forward=forward, maximum=maximum,
                                     check=check, _here_first=_here_first)

    def determine_ntrack(self, maximum=None):
        """Try 16, 32 and 64 tracks (mark4/base.py:168-207)."""
        old_ntrack = self.ntrack
        for ntrack in (16, 32, 64):
            self.ntrack = ntrack
            with self.temporary_offset():
                offsets = self.locate_frames(maximum=maximum)
            if offsets:
                self.fh_raw.seek(offsets[0])
                return ntrack
        self.ntrack = old_ntrack
        raise HeaderNotFoundError("cannot determine ntrack automatically. "
                                  "(tried 16, 32, 64). Try passing in an "
                                  "explicit value.")

    def get_frame_rate(self):
        """From the time step between the first two frames
        (mark4/base.py:87-108)."""
        with self.temporary_offset(0):
            header0 = self.find_header()
            self.fh_raw.seek(header0.frame_nbytes, 1)
            header1 = self.read_header()
        tdelta = (header1.fraction[0] - header0.fraction[0]) % 1.
        return float(np.round(1. / tdelta))


class Mark4FileWriter(FileBase):
    """Frame-level writer (mark4/base.py:210-232)."""

    def write_frame(self, data, header=None, **kwargs):
        if not isinstance(data, Mark4Frame):
            if header is None:
                header = Mark4Header.fromvalues(**kwargs)
            data = Mark4Frame.fromdata(data, header)
        return data.tofile(self.fh_raw)


class Mark4StreamReader(GPUStreamReaderBase):
    """Mark 4 stream -> device tensor (nsample, nchan)."""
    _sample_shape_fields = ('nchan',)

    def __init__(self, fh_raw, sample_rate=None, ntrack=None, decade=None, ref_time=None,
                 squeeze=True, subset=(), fill_value=0., verify='fix'):
        # the header's year is a single digit: something has to pin the decade
        if ref_time is None and decade is None:
            raise TypeError("reading Mark 4 needs `decade` or `ref_time` to "
                            "resolve the single-digit year of the headers.")
        fh_raw = Mark4FileReader(fh_raw, ntrack=ntrack, decade=decade, ref_time=ref_time)
        header0 = fh_raw.find_header()
        offset0 = fh_raw.tell()
        if sample_rate is None:
            sample_rate = fh_raw.get_frame_rate() * header0.samples_per_frame
        super().__init__(
            fh_raw, header0, sample_rate=hz(sample_rate),
            samples_per_frame=header0.samples_per_frame,
            unsliced_shape=(header0.nchan,), bps=header0.bps,
            complex_data=False, squeeze=squeeze, subset=subset,
            fill_value=fill_value, verify=verify)
        self._ntrack = header0.ntrack
        self._set_nbytes = header0.frame_nbytes
        self._file_offset0 = offset0
        frame_rate = self.sample_rate / self.samples_per_frame
        self._frame_qms = int(round(4000. / frame_rate))
        self._coder = (header0.nchan, header0.magnitude_signature() or header0.bps,
                       header0.fanout)
        self._start_time = header0.get_time()
        self._ref_qms = header0.time_quarter_ms()
        # a subset that picks channels becomes shorter bit maps for the decode
        # kernel (bb_decode_mark4_select) instead of an indexing pass afterwards
        self._plan_channel_select(self.subset)
        if self._within_np is not None and header0.fanout * len(self._within_np) > 32:
            self._within_np, self._decode_shape = None, self._unsliced_shape

    def _count_samples(self):
        last, header0 = self._last_header, self.header0
        dq = last.time_quarter_ms() - self._ref_qms
        if last.year != header0.year:
            y = header0.year
            leap = (y % 4 == 0 and (y % 100 != 0 or y % 400 == 0))
            dq += (365 + leap) * 86400 * 4000
        return (int(round(dq / self._frame_qms)) + 1) * self.samples_per_frame

    def _image(self):
        return self.fh_raw.image()

    def _find_last_header(self):
        """Last frame of the file: searched backwards from one frame before
        the end, with a sync pattern required one frame earlier and (if inside
        the file) one later (base/base.py:1066-1077; mark4/base.py:307-314)."""
        fn = self._set_nbytes
        size = len(self._image())
        with self.fh_raw.temporary_offset(max(0, size - fn)):
            try:
                header = self.fh_raw.find_header(forward=False, maximum=2 * fn - 1,
                                                 check=(-1, 1))
            except HeaderNotFoundError as exc:
                exc.args += ("corrupt VLBI frame? No frame in last {0} bytes."
                             .format(2 * fn),)
                raise
        header.infer_decade(self.start_time)
        return header

    # -- corruption-tolerant index
    _can_relocate = True
    _resident = None

    def _relocate(self):
        """Frames are missing or out of place: keep the file in HBM, find every
        intact frame byte by byte (bb_mark4_locate), read those headers
        (bb_mark4_scan_at) and place the frames by their time index; frames
        without an entry decode to fill_value -- the outcome of the
        reference's _bad_frame recovery (base/base.py:1127-1219).  A frame
        that shows up more than once is 'excess data', which the reference
        refuses as well."""
        kernels.require_gpu()
        image = self._image()
        dev, n = self._whole_file_in_hbm(), len(image)
        offs = kernels.mark4_locate(dev, n, self._ntrack)
        offs = self._of_this_stream(dev, n, offs)
        recs = kernels.mark4_scan_at(dev, n, offs, self._ntrack, self.header0.year,
                                     self._ref_qms, self._frame_qms)
        ok = ((recs[:, 3] >> 16) & _lib.FRAME_OK) != 0
        index = recs[:, 2][ok]
        if index.numel() != torch.unique(index).numel():
            raise AssertionError("problem loading frame: there appears to be excess data "
                                 "(a frame time occurs more than once).")
        nsets = self._nsample // self.samples_per_frame
        self._resident = (dev, kernels.build_index(recs, nsets, 1, None))
        self._located = (offs, recs)
        self._relocated = True
        self._note_damage(self._resident[1])

    def _of_this_stream(self, dev, n, offs):
        """Of the located frames (sync pattern here and one frame later) those whose headers
        also carry what every header of THIS stream carries -- head stacks, track roll, system id
        (mark4/header.py:144-154) -- here and, where it fits, one frame later: the reader's
        searches hand header0 to locate_frames (base/base.py:1127-1219), which compares all of
        its invariant bits at the candidate and at the check positions."""
        if not offs.numel():
            return offs
        pattern, mask = self.header0.invariant_pattern()
        pb = np.ascontiguousarray(pattern).view(np.uint8)
        mb = np.ascontiguousarray(mask).view(np.uint8)
        used = np.nonzero(mb)[0]
        lo, hi = int(used[0]), int(used[-1]) + 1
        pb = torch.from_numpy(pb[lo:hi].copy()).to(dev.device)
        mb = torch.from_numpy(mb[lo:hi].copy()).to(dev.device)
        span = torch.arange(lo, hi, device=dev.device)
        flat = dev.reshape(-1).view(torch.uint8)

        def agrees(pos):
            out = torch.empty(pos.numel(), dtype=torch.bool, device=dev.device)
            for a in range(0, pos.numel(), 1 << 14):            # (bounded gathers)
                at = pos[a:a + (1 << 14), None] + span
                out[a:a + (1 << 14)] = (((flat[at.clamp(max=flat.numel() - 1)] ^ pb) & mb) == 0).all(dim=1)
            return out
        o64 = offs.to(torch.int64)
        after = o64 + self._set_nbytes
        fits = after + hi <= n
        keep = agrees(o64) & (~fits | agrees(torch.where(fits, after, o64)))
        return offs[keep]

    def _maps(self):
        """(sign bits, magnitude bits, selected?) for the decode kernel."""
        maps = BITMAPS[self._coder]          # KeyError: unsupported Mark 4 mode
        sign, mag = maps['sign_bit'], maps['mag_bit']
        if self._within_np is None:
            return sign, mag, False
        return kernels.mark4_select_maps(sign, mag, self._unsliced_shape[0],
                                         self._within_np) + (True,)

    def _read_sets(self, first, last, into=None):
        if self._resident is None:
            return super()._read_sets(first, last, into)
        dev, src = self._resident
        self._warn_damage(first, last)
        sign, mag, select = self._maps()
        flat = kernels.decode_mark4(
            dev, last - first, self._ntrack, 20000, sign, mag,
            fill_words=160, src=src[first:last].contiguous(), fill_value=self.fill_value,
            out=into, select=select)
        return flat.reshape(((last - first) * self.samples_per_frame,)
                            + tuple(self._decode_shape))

    def header_crc_errors(self):
        """Longitudinal check of every frame header: int64 device tensor, one
        entry per frame of the file, bit t set when track t's 160 header bits
        fail their CRC-12 (``baseband.mark4.header.crc12.check`` in the
        reference, mark4/header.py:34-44; the reference does not apply it
        while reading and neither does `read`: decoded samples are never
        changed by this report).  The file is looked at in HBM
        (bb_mark4_header_crc)."""
        dev = self._whole_file_in_hbm()
        nframes = (len(self._image()) - self._file_offset0) // self._set_nbytes
        if self._resident is not None:              # frames were located individually
            offs = self._resident[1]
            offs = offs[offs >= 0].contiguous()
            return kernels.mark4_header_crc(dev, offs.numel(), self._ntrack, offsets=offs)
        return kernels.mark4_header_crc(dev, nframes, self._ntrack, first_offset=self._file_offset0)

    _window = None          # kernels.Mark4Window: argument blocks of the one-call window

    def _process_window(self, dbuf, first, last, out_flat):
        """scan -> index -> verification -> decode of frames [first, last): one
        library call (bb_mark4_read_window)."""
        n = last - first
        # one header beyond the request is checked too when it was staged
        nframes = min(n + (1 if self.verify else 0), dbuf.numel() // self._set_nbytes)
        w = self._window
        if w is None:
            sign, mag, select = self._maps()
            w = self._window = kernels.Mark4Window(self._ntrack, self.header0.year, self._ref_qms,
                                                   self._frame_qms, 20000, sign, mag, select, 160,
                                                   self.fill_value)
        if w.fill_value != self.fill_value:
            w.set_fill(self.fill_value)
        nbad = verified = None
        if self.verify:
            # (queued before the decode, an event behind it: `_resolve_checks`
            # waits for this verdict alone)
            nbad, verified = self._verdict_targets()
        # the look-ahead header (record n) only has to be a header
        w.scan.by_position = 0 if self.verify else 1        # (verify=False: frames by position, nothing checked)
        w.run(dbuf, first, nframes, n, out_flat, min(n, nframes), nbad, verified, scan_stream=self._scan_side)
        if self.verify:
            self._note_checked(nframes, missing=max(0, n - nframes))


class Mark4StreamWriter(GPUStreamWriterBase):
    """Mark 4 stream writer (mark4/base.py:315-334): (n, nchan) samples are
    track-multiplexed on the GPU; the first 160*fanout samples of every frame
    are dropped because the headers occupy their place on tape."""
    _sample_shape_fields = ('nchan',)

    def __init__(self, fh_raw, header0=None, sample_rate=None, squeeze=True,
                 time=None, ntrack=64, bps=2, fanout=4, **kwargs):
        if header0 is None:
            # (further keywords -- nchan / sample_shape, decade, ... -- are the header's,
            # as in the reference, whose opener hands all of them to fromvalues)
            header0 = Mark4Header.fromvalues(ntrack, time=time, bps=bps, fanout=fanout, **kwargs)
        elif kwargs:
            raise TypeError("unexpected keyword(s) {} next to header0".format(sorted(kwargs)))
        if sample_rate is None:
            raise ValueError("Mark 4 stream writer needs a sample_rate.")
        super().__init__(fh_raw, header0, sample_rate=sample_rate,
                         samples_per_frame=header0.samples_per_frame,
                         unsliced_shape=(header0.nchan,), bps=header0.bps,
                         complex_data=False, squeeze=squeeze)
        self._frame_rate = self.sample_rate / self.samples_per_frame
        self._start_time = header0.get_time()
        self._coder = (header0.nchan, header0.magnitude_signature() or header0.bps,
                       header0.fanout)

    def _write_frames(self, data, valid):
        h0 = self.header0
        maps = BITMAPS[self._coder]
        spf = self.samples_per_frame
        nfr = data.shape[0] // spf
        nfill = 160 * h0.fanout
        body = data.reshape(nfr, spf, h0.nchan)[:, nfill:].contiguous()
        words = kernels.encode_mark4(body, h0.ntrack, maps['sign_bit'], maps['mag_bit'])
        k = np.arange(self._nframes_written, self._nframes_written + nfr)
        times = self._start_time + np.rint(k * 1e9 / self._frame_rate).astype('m8[ns]')
        invalid = ~np.asarray(valid, bool)
        heads = frame_header_streams(h0, times, invalid=invalid, before_invalid=self._before_invalid)
        if nfr:
            self._before_invalid = bool(invalid[-1])
        self._emit_frames(heads.view(np.uint8).reshape(nfr, -1), words)

    _before_invalid = False     # whether the frame written last was flagged invalid (its flag is in the next CRC)


def _adopt_header(h):
    """The reference's Mark4Header -> ours (same (5, ntrack) words, decade)."""
    if isinstance(h, Mark4Header) or not hasattr(h, 'words'):
        return h
    return Mark4Header(np.array(h.words, dtype=np.uint32), decade=getattr(h, 'decade', None), verify=False).copy()


open = FormatOpener('Mark4', {'rb': Mark4FileReader, 'wb': Mark4FileWriter,
                              'rs': Mark4StreamReader,
                              'ws': Mark4StreamWriter}, adopt_header=_adopt_header)
open.__doc__ = """Open Mark 4 file(s): ``'rb'`` -> `Mark4FileReader`, ``'rs'`` ->
`Mark4StreamReader`, ``'ws'`` -> `Mark4StreamWriter` (mark4/base.py:337-430);
names, handles, lists of names and ``{file_nr}`` templates are accepted."""
