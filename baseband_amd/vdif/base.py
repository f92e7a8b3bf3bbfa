"""VDIF file and stream readers, and ``open``.

Mirrors ``VDIFFileReader`` (vdif/base.py:70-298), ``VDIFStreamReader``
(vdif/base.py:413-755) and ``open`` (vdif/base.py:810-884) for the reading
modes ``'rb'`` and ``'rs'``.  The stream reader's per-frame-set Python loop
(base/base.py:957-967 -> vdif/frame.py:176-243,402-434) is replaced per staged
window by three launches: ``bb_vdif_scan`` -> ``bb_build_index`` ->
``bb_decode_frames``.
"""

import warnings

import numpy as np
import torch

from .. import _lib, kernels
from ..base.base import (FileBase, VLBIFileReaderBase, GPUStreamReaderBase,
                         HeaderNotFoundError)
from ..base.writer import GPUStreamWriterBase
from ..base.opener import FormatOpener
from ..base.header import strided_header_words
from .header import VDIFHeader, frame_header_words
from .frame import VDIFFrame, VDIFFrameSet
from ..base.quantities import hz

__all__ = ['VDIFFileReader', 'VDIFFileWriter', 'VDIFStreamReader', 'VDIFStreamWriter',
           'open']


class VDIFFileReader(VLBIFileReaderBase):
    """Simple reader for VDIF files: headers, frames, frame sets."""
    _format = 'vdif'
    _info_find_kwargs = {'maximum': 0}      # only at the start (vdif/file_info.py:24-30)

    def _info_extras(self, header0, offset0):
        """edv, thread ids and number of frame sets (vdif/file_info.py:10-50)."""
        with self.temporary_offset(0):
            thread_ids = self.get_thread_ids()
        nframes = len(self.image()) / header0.frame_nbytes
        nsets = nframes / len(thread_ids)
        # (a file that is not a whole number of FRAMES is reported as that alone: vdif/file_info.py:45-55)
        notes = {} if nsets % 1 == 0 or nframes % 1 else {
            'number_of_framesets': 'file contains non-integer number ({}) of framesets'.format(nsets)}
        return {'edv': header0.edv, 'thread_ids': thread_ids,
                'number_of_framesets': int(nsets) if nsets % 1 == 0 else None,
                '_sample_shape': (len(thread_ids), header0.nchan), '_warnings': notes}

    def read_header(self, edv=None, verify=True):
        return VDIFHeader.fromfile(self.fh_raw, edv=edv, verify=verify)

    def read_frame(self, edv=None, verify=True):
        return self._lend_device_words(VDIFFrame.fromfile(self.fh_raw, edv=edv, verify=verify))

    def read_frameset(self, thread_ids=None, edv=None, verify=True):
        return VDIFFrameSet.fromfile(self.fh_raw, thread_ids, edv=edv,
                                     verify=verify)

    def find_header(self, pattern=None, *, edv=None, mask=None, frame_nbytes=None,
                    offset=0, forward=True, maximum=None, check=1):
        """Nearest header from the current position (vdif/base.py:216-316).

        With a `pattern` (normally a header of the stream: its invariant bits
        and frame size are used) this is a masked byte search
        (`locate_frames`).  Without one, a header is read at every position --
        almost any bytes parse as a VDIF header -- and accepted when the same
        stream invariants are found `check` frames away."""
        if pattern is not None:
            kw = dict(mask=mask, frame_nbytes=frame_nbytes, offset=offset, forward=forward,
                      maximum=maximum, check=check)
            # (the current position first: it is the nearest candidate there can be)
            locations = (self.locate_frames(pattern, _here_first=True, **kw)
                         or self.locate_frames(pattern, **kw))
            if not locations:
                raise HeaderNotFoundError('could not locate a a nearby frame.')
            self.fh_raw.seek(locations[0])
            with self.temporary_offset():
                return self.read_header(edv=getattr(pattern, 'edv', None))
        if maximum is None:
            maximum = 10000 if frame_nbytes is None else 2 * frame_nbytes
        here = self.fh_raw.tell()
        positions = (range(here, here + maximum + 1) if forward
                     else range(here, max(here - maximum - 1, -1), -1))
        for pos in positions:
            self.fh_raw.seek(pos)
            try:
                header = self.read_header(edv=edv)
            except Exception:
                continue
            if frame_nbytes is not None and frame_nbytes != header.frame_nbytes:
                continue
            self.fh_raw.seek(pos)
            try:
                return self.find_header(header, maximum=0, check=check)
            except Exception:
                continue
        self.fh_raw.seek(here)
        raise HeaderNotFoundError("could not locate a nearby header.")

    def _header_table(self, header0, offset=0):
        """(nframes, nwords) view of all headers at the fixed frame stride."""
        return strided_header_words(self.image(), header0.frame_nbytes,
                                    header0.nbytes // 4, offset=offset)

    def get_frame_rate(self):
        """Frames per second: largest frame number seen within the first
        second, plus one (base/base.py:371-406); falls back to the EDV 1/3
        header sample rate (vdif/base.py:150-170)."""
        with self.temporary_offset(0):
            header0 = self.read_header()
        hw = self._header_table(header0)
        # the table is a strided view of the whole (mapped) file: look at a
        # growing prefix, so that a multi-GiB file is not paged in to find the
        # first wrap of the frame counter
        m = min(len(hw), 4096)
        while True:
            frame_nr = hw[:m, 1] & 0xffffff
            # skip the first frame number, then walk until the count wraps to 0
            differ = np.nonzero(frame_nr != frame_nr[0])[0]
            if len(differ):
                i = differ[0]
                wrap = np.nonzero(frame_nr[i:] == 0)[0]
                if len(wrap):
                    j = i + wrap[0]
                    return int(max(frame_nr[0], frame_nr[i:j].max() if j > i else 0)) + 1
            if m == len(hw):
                break
            m = min(len(hw), m * 8)
        rate = header0.frame_rate
        if rate is None:
            raise EOFError("file contains less than one second of data and "
                           "the header does not provide a sample rate.")
        return int(round(rate))

    def get_thread_ids(self, check=2):
        """Sorted thread ids: frame sets are scanned until the set of ids
        stops growing for `check` sets (vdif/base.py:172-215)."""
        pos = self.fh_raw.tell()
        with self.temporary_offset():
            header0 = self.read_header()
        hw = self._header_table(header0, offset=pos)
        n = len(hw)
        m = min(n, 256)                     # growing prefix of the strided view
        while True:
            frame_nrs = hw[:m, 1] & 0xffffff
            threads = (hw[:m, 3] >> 16) & 0x3ff
            seen, n_check, k = set(), 1, 0
            while n_check > 0 and k < m:
                fnr, n0 = frame_nrs[k], len(seen)
                while k < m and frame_nrs[k] == fnr:
                    seen.add(int(threads[k]))
                    k += 1
                n_check = check if len(seen) > n0 else n_check - 1
            if n_check <= 0 and k < m:
                break                       # decided inside the prefix
            if m == n:
                # ran off the end: very short files (like the samples) are let through
                if n_check > 0 and len(self.image()) - pos > check * len(seen) * header0.frame_nbytes:
                    raise EOFError
                break
            m = min(n, m * 8)
        return sorted(seen)


class VDIFFileWriter(FileBase):
    """Frame-level writer (vdif/base.py:318-363): samples are packed on the
    GPU by ``VDIFFrame.fromdata`` / ``VDIFFrameSet.fromdata``."""

    def write_frame(self, data, header=None, **kwargs):
        if not isinstance(data, VDIFFrame):
            data = VDIFFrame.fromdata(data, header, **kwargs)
        return data.tofile(self.fh_raw)

    def write_frameset(self, data, header=None, **kwargs):
        if not isinstance(data, VDIFFrameSet):
            data = VDIFFrameSet.fromdata(data, header, **kwargs)
        return data.tofile(self.fh_raw)


class VDIFStreamReader(GPUStreamReaderBase):
    """VDIF stream -> device tensor of shape (nsample, nthread, nchan)
    (squeezed / subset as requested).

    Parameters are those of the reference reader (vdif/base.py:413-440);
    ``sample_rate`` is a plain number in Hz.
    """
    _sample_shape_fields = ('nthread', 'nchan')

    def __init__(self, fh_raw, sample_rate=None, squeeze=True, subset=(), fill_value=0.,
                 verify='fix'):
        fh_raw = VDIFFileReader(fh_raw)
        header0 = fh_raw.read_header()
        fh_raw.seek(0)
        self._file_threads = thread_ids = fh_raw.get_thread_ids()
        nthread = len(thread_ids)
        if sample_rate is None:
            sample_rate = header0.sample_rate
            if sample_rate is None:
                sample_rate = fh_raw.get_frame_rate() * header0.samples_per_frame
        sample_rate = hz(sample_rate)
        super().__init__(
            fh_raw, header0, sample_rate=sample_rate,
            samples_per_frame=header0.samples_per_frame,
            unsliced_shape=(nthread, header0.nchan), bps=header0.bps,
            complex_data=header0.complex_data, squeeze=squeeze, subset=subset,
            fill_value=fill_value, verify=verify)
        self._frame_rate = int(round(sample_rate / self.samples_per_frame))
        self._frame_nbytes = header0.frame_nbytes
        self._set_nbytes = header0.frame_nbytes * nthread
        self._file_offset0 = 0
        # thread part of the subset is applied while reading
        # (vdif/base.py:464-490)
        self._thread_ids, self._frameset_subset = thread_ids, self.subset
        if self.subset and (nthread > 1 or not self.squeeze):
            picked = np.array(thread_ids)[self.subset[0]]
            self._thread_ids = np.atleast_1d(picked.squeeze()).tolist()
            # what is left to do on the thread axis of a decoded frame set,
            # which holds only the picked threads
            if picked.ndim == 0:
                lead = () if self.squeeze else (0,)
            elif self.squeeze and len(self._thread_ids) == 1:
                lead = (np.newaxis,)
            else:
                lead = (slice(None),)
            self._frameset_subset = lead + self.subset[1:]
        self._decode_shape = (len(self._thread_ids), header0.nchan)
        self._thread_slot = None
        self._pattern, self._mask = header0.invariant_pattern()
        self._start_time = header0.get_time(frame_rate=self._frame_rate)
        self._coder = (_lib.CODER_MARK5B if header0.edv == 0xab
                       else _lib.CODER_VDIF)
        self._resident = None
        self._plan_channel_select(self._frameset_subset, payload_nbytes=header0.payload_nbytes)
        # (the number of samples follows from the last header, looked for when it is
        # first asked for -- `_nsample`, `_last_header` -- as the reference's lazy
        # property does: a file whose tail holds no frame of header0's thread opens,
        # and raises HeaderNotFoundError where its length is needed, vdif/base.py:492-517)

    def _count_samples(self):
        return (self._get_index(self._last_header) + 1) * self.samples_per_frame

    _can_relocate = True

    def _image(self):
        return self.fh_raw.image()

    def _lost_behind_holes(self, dev, offs, recs, nbytes):
        """The reference assembles a frame set frame by frame and, when a frame
        cannot be read, looks for the next header no further than two frames
        on; if there is none it gives up on the REST of that set
        (vdif/base.py:655-690).  Same here: a hole of more than three frame
        lengths between two located frames of one set drops the later ones."""
        h0 = self.header0
        if offs.numel() > 1:
            t = recs[:, 2]
            same_set = t[1:] == t[:-1]
            hole = (offs[1:] - offs[:-1]) > 3 * self._frame_nbytes + h0.nbytes - 1
            brk = torch.zeros_like(t)
            brk[1:] = (hole & same_set).to(t.dtype)
            if bool(brk.any()):
                c = torch.cumsum(brk, 0)
                start = torch.zeros_like(t, dtype=torch.bool)
                start[0] = True
                start[1:] = ~same_set
                idx = torch.arange(t.numel(), device=t.device)
                first = torch.cummax(torch.where(start, idx, torch.zeros_like(idx)), 0).values
                lost = (c - c[first]) > 0
                recs[:, 3] = torch.where(lost, recs[:, 3] & ~(_lib.FRAME_OK << 16), recs[:, 3])
        return recs

    def _lost_behind_foreign_headers(self, dev, offs, recs, nbytes):
        """The reference collects the frames of a set one after the other (vdif/base.py:655-712).
        When a frame cannot be read it searches for the next header from the end of the
        header before; a header found that way -- not where the next frame was due -- ends the
        set when its frame number is another one; frames of the set that follow it are not
        used.  A header that stands where it is due ends the set when it can be read as a
        frame of another set, and is stepped over when its seconds make no sense (its read
        fails, the search from there finds the header after it)."""
        h0 = self.header0
        if offs.numel() > 2:
            t = recs[:, 2].to(torch.int64)
            n = t.numel()
            o64 = offs.to(torch.int64)
            run_start = torch.ones(n, dtype=torch.bool, device=t.device)
            run_start[1:] = t[1:] != t[:-1]
            run_id = torch.cumsum(run_start.to(torch.int64), 0) - 1
            top = nbytes // self._set_nbytes + 2
            valid = (t >= 0) & (t < top)
            first_run = torch.full((top,), n, dtype=torch.int64, device=t.device)
            first_run.scatter_reduce_(0, t[valid], run_id[valid], 'amin')
            resumed = valid & (run_id > first_run[t.clamp(0, top - 1)])
            if bool(resumed.any()):
                starts = torch.nonzero(run_start)[:, 0]
                ends = torch.cat([starts[1:], torch.tensor([n], device=t.device)])
                before = (run_id - 1).clamp(min=0)
                p_ = starts[before]                                  # first record of the run in between
                single = (ends[before] - p_) == 1
                due = o64[p_] == o64[(p_ - 1).clamp(min=0)] + self._frame_nbytes
                apart = t[p_] - t
                senseless = (apart <= -self._frame_rate) | (apart >= 2 * self._frame_rate)
                word1 = dev[(o64 + 4)[:, None] + torch.arange(4, device=dev.device)].to(torch.int64)
                frame_nr = (word1[:, 0] | (word1[:, 1] << 8) | (word1[:, 2] << 16))
                same_nr = frame_nr[p_] == frame_nr
                lost = resumed & ~(single & ((~due & same_nr) | (due & senseless)))
                recs[:, 3] = torch.where(lost, recs[:, 3] & ~(_lib.FRAME_OK << 16), recs[:, 3])
        return recs

    def _lost_twice_or_in_front(self, dev, offs, recs, nbytes):
        """Two more of the reference's habits.  A thread that shows up twice in one set is
        discarded, both times (vdif/base.py:700-705).  And a damaged set is taken from the
        first header that has a header one frame before it (the backward search for the start
        of the set insists on that, vdif/base.py:576-612): frames of the set in front of it
        whose predecessor is gone are not used -- unless the set before was damaged as well."""
        h0 = self.header0
        if offs.numel() > 1:
            t = recs[:, 2].to(torch.int64)
            n = t.numel()
            o64 = offs.to(torch.int64)
            top = nbytes // self._set_nbytes + 2
            live = (((recs[:, 3] >> 16) & _lib.FRAME_OK) != 0) & (t >= 0) & (t < top)
            tc = t.clamp(0, top - 1)
            key = tc * 1024 + (recs[:, 3] & 0x3ff).to(torch.int64)
            twice = torch.zeros(n, dtype=torch.bool, device=t.device)
            if bool(live.any()):
                _, back, times = torch.unique(key[live], return_inverse=True, return_counts=True)
                twice[live] = times[back] > 1
            per_set = torch.zeros(top, dtype=torch.int64, device=t.device)
            per_set.scatter_add_(0, tc[live & ~twice], torch.ones_like(tc[live & ~twice]))
            damaged = per_set[tc] < len(self._file_threads)
            at = torch.searchsorted(o64, o64 - self._frame_nbytes).clamp(max=n - 1)
            follows = (o64[at] == o64 - self._frame_nbytes) | (o64 < self._file_offset0 + self._frame_nbytes)
            order = torch.argsort(tc, stable=True)
            f_sorted = (follows & live)[order].to(torch.int64)
            cs = torch.cumsum(f_sorted, 0)
            g = tc[order]
            head = torch.ones(n, dtype=torch.bool, device=t.device)
            head[1:] = g[1:] != g[:-1]
            base = torch.cummax(torch.where(head, cs - f_sorted, torch.zeros_like(cs)), 0).values
            none_before = torch.empty(n, dtype=torch.bool, device=t.device)
            none_before[order] = (cs - f_sorted - base) == 0
            # (when the set before was damaged too, the reference arrives at this one by the
            # header it ran into at the end of that one, and takes it from there as it is)
            before_whole = torch.ones(top, dtype=torch.bool, device=t.device)
            before_whole[1:] = per_set[:-1] >= len(self._file_threads)
            front = live & damaged & before_whole[tc] & ~follows & none_before
            lost = twice | front
            recs[:, 3] = torch.where(lost, recs[:, 3] & ~(_lib.FRAME_OK << 16), recs[:, 3])
        return recs

    def _relocate(self):
        """Corruption-tolerant index: keep the file resident in
        HBM, find every intact frame with the byte-granular header search
        (bb_vdif_locate: pattern + a header one frame later), scan those
        headers and place the frames by (time index, thread).  Frames that
        are missing or damaged simply have no entry and decode to fill_value
        -- what the reference's _bad_frame achieves frame set by frame set
        (vdif/base.py:536-755)."""
        kernels.require_gpu()
        h0 = self.header0
        image = self._image()
        dev = self._whole_file_in_hbm()         # what earlier windows left in HBM is not sent again
        n = len(image)
        offs = kernels.vdif_locate(dev, n, self._frame_nbytes, h0.nbytes,
                                   self._pattern, self._mask)
        recs = kernels.vdif_scan_at(dev, n, offs, self._frame_nbytes, h0.nbytes,
                                    self._pattern, self._mask, h0['seconds'],
                                    h0['frame_nr'], self._frame_rate)
        if self._thread_slot is None:
            self._thread_slot = kernels.thread_slot_map(self._thread_ids, dev.device)
        # three rules of the reference's frame-by-frame recovery decide which located frames count
        recs = self._lost_behind_holes(dev, offs, recs, n)
        recs = self._lost_behind_foreign_headers(dev, offs, recs, n)
        recs = self._lost_twice_or_in_front(dev, offs, recs, n)
        ok = ((recs[:, 3] >> 16) & _lib.FRAME_OK) != 0
        same = (recs[:, 3] & 0xffff) == h0['thread_id']
        sel = recs[:, 2][ok & same]
        nsets = int(sel.max().item()) + 1 if sel.numel() else 0
        src = kernels.build_index(recs, nsets, len(self._thread_ids), self._thread_slot)
        self._resident = (dev, src)
        self._nsample = nsets * self.samples_per_frame
        self._located = (offs, recs)
        self._relocated = True
        self._note_damage(src, len(self._thread_ids))
        self._note_sets_before_trouble(offs, recs)

    _before_trouble = frozenset()   # whole sets whose successor cannot be read from where they end

    def _note_sets_before_trouble(self, offs, recs, most=256):
        """The reference reads one frame set AHEAD of the one it returns (base/base.py:1115-1123);
        when that read fails, and the header it started at is not of the next set, the set being
        returned -- whole, and where it should be -- still gets a bare "problem loading frame set
        k." (vdif/base.py:547-562).  That is the case for the whole set in front of a run of
        damaged ones when the frames behind it begin inside a set, or not with a header."""
        self._before_trouble = frozenset()
        sets = self._damage[0] if self._damage is not None else ()
        if not len(sets) or len(self._thread_ids) != len(self._file_threads):
            return
        run_starts = [int(d) for i, d in enumerate(sets) if d > 0 and (i == 0 or sets[i - 1] != d - 1)][:most]
        if not run_starts:
            return
        from .. import _lib
        t = recs[:, 2].to(torch.int64)
        ok = ((recs[:, 3] >> 16) & _lib.FRAME_OK) != 0
        o64 = offs.to(torch.int64)
        thread = (recs[:, 3] & 0x3ff).to(torch.int64)
        wanted = set(int(x) for x in self._thread_ids)
        found = set()
        for d in run_starts:
            k = d - 1
            mine = torch.nonzero(ok & (t == k))[:, 0]
            if not mine.numel():
                continue
            end = int(o64[mine].max()) + self._frame_nbytes
            q = int(torch.searchsorted(o64, torch.tensor([end], device=o64.device))[0])
            if q >= o64.numel():
                continue                                    # (the file ends there: the last set is taken as it is)
            if int(o64[q]) != end or not bool(ok[q]):
                found.add(k)                                # (no header where the next set should begin)
                continue
            j = int(t[q])
            if j == k + 1:
                continue
            stop = q
            while stop < o64.numel() and int(t[stop]) == j and int(o64[stop]) == end + (stop - q) * self._frame_nbytes:
                stop += 1
            if not wanted <= set(int(x) for x in thread[q:stop].tolist()):
                found.add(k)
        self._before_trouble = frozenset(found)

    def _warn_damage(self, first, last):
        if self.verify is not True:
            for k in sorted(self._before_trouble):
                if first <= k < last:
                    warnings.warn("problem loading frame set {}.".format(k))
        super()._warn_damage(first, last)

    def _verification_error(self, msg):
        """The reference loads one frame set at a time from the fixed stride and stops at the first
        it cannot complete: the file ends inside the set (EOFError), the next set begins before every
        requested thread was seen (OSError, vdif/frame.py:436-475), or a complete set sits where
        another was expected (ValueError, vdif/base.py:640-652).  The same walk over the header
        table says which of the three this read would have met."""
        try:
            kind = self._first_problem_met()
        except Exception:               # (a table that cannot be read as headers: the general answer)
            kind = 'number'
        return self._strict_error(kind, msg)

    def _strict_error(self, kind, msg):
        if kind is None:
            return None
        if kind == 'end':
            return EOFError("the file ends inside a frame set. " + msg)
        if kind == 'header':            # (bytes went missing: what stands at a frame boundary is no header)
            return AssertionError("a header failed verification. " + msg)
        if kind == 'threads':
            return OSError("could not find all requested frames. " + msg)
        return ValueError("wrong frame number. " + msg)

    def _first_problem_met(self, first=None, last=None):
        """Walks the frame sets [first, last) (default: those of the read in progress) as the
        reference's `VDIFFrameSet.fromfile` would from the fixed stride (vdif/frame.py:201-235) and
        names the first problem: 'end' (the file ends inside a set), 'header' (bytes that are no
        header where one is due), 'threads' (the next set begins, or a thread repeats, before all
        threads asked for were seen), 'number' (a whole set of another time), or None."""
        if first is None:
            if self._asked is None:
                return None
            spf = self.samples_per_frame
            first = self._asked[0] // spf
            last = -(-(self._asked[0] + self._asked[1]) // spf)
        hw = self.fh_raw._header_table(self.header0, offset=self._file_offset0)
        per_set, nfr = len(self._file_threads), len(hw)
        nfit = (len(self._image()) - self._file_offset0) // self._frame_nbytes    # frames that are there whole
        wanted = set(int(t) for t in self._thread_ids)

        def is_header(words):
            return not any(((int(w) ^ p_) & m_) for w, p_, m_ in zip(words, self._pattern, self._mask))
        # (sets that stand whole on the fixed stride are skipped in one NumPy pass: the walk
        # below starts at the first one that does not)
        whole = min(last, nfit // per_set) - first
        if whole > 0:
            block = np.asarray(hw[first * per_set:(first + whole) * per_set]).reshape(whole, per_set, -1)
            pat = np.asarray(self._pattern, dtype=np.uint32)
            msk = np.asarray(self._mask, dtype=np.uint32)
            sound = (((block ^ pat) & msk) == 0).all(axis=(1, 2))
            sec = (block[:, :, 0] & 0x3fffffff).astype(np.int64)
            nr = (block[:, :, 1] & 0xffffff).astype(np.int64)
            index = (sec - self.header0['seconds']) * self._frame_rate + nr - self.header0['frame_nr']
            sound &= (index == (first + np.arange(whole))[:, None]).all(axis=1)
            thread = (block[:, :, 3] >> 16) & 0x3ff
            for t_ in wanted:
                sound &= (thread == t_).any(axis=1)
            odd = np.nonzero(~sound)[0]
            first = first + (int(odd[0]) if len(odd) else whole)
        for k in range(first, last):
            p = k * per_set
            if p >= nfr:
                return 'end'
            if not is_header(hw[p]):
                return 'header'
            when = (int(hw[p, 0]) & 0x3fffffff, int(hw[p, 1]) & 0xffffff)
            seen, j = set(), p
            while j < nfr:
                if not is_header(hw[j]):
                    return 'header'
                if (int(hw[j, 0]) & 0x3fffffff, int(hw[j, 1]) & 0xffffff) != when:
                    break
                thread = (int(hw[j, 3]) >> 16) & 0x3ff
                if thread in wanted and thread in seen:     # (a thread again: the set is over, vdif/frame.py:209)
                    break
                if thread in wanted and j >= nfit:          # (its payload is cut short)
                    return 'end'
                seen.add(thread)
                j += 1
                if wanted <= seen:
                    break
            if not wanted <= seen:
                return 'end' if j >= nfr else 'threads'
            index = (when[0] - self.header0['seconds']) * self._frame_rate + when[1] - self.header0['frame_nr']
            if index != k:
                return 'number'
        return None

    def _damage_message(self, k, missing):
        # (the reference's two sentences, vdif/base.py:700-730)
        if missing.all():
            return ("problem loading frame set {}. The frame set seems to be missing altogether. "
                    "All threads set to invalid.".format(k))
        ids = [int(t) for t, gone in zip(self._thread_ids, missing) if gone]
        return "problem loading frame set {}. Thread(s) {} missing; set to invalid.".format(k, ids)

    def _slip_message(self, k, nbytes):
        return "problem loading frame set {}. Stream off by {} bytes.".format(k, nbytes)

    def _read_sets(self, first, last, into=None):
        if self._resident is None:
            return super()._read_sets(first, last, into)
        dev, src = self._resident
        if self.verify is True:         # (strict: every read answers for the sets it covers)
            err = self._strict_error(self._first_problem_met(first, last),
                                     "problem loading frame set in {}..{}".format(first, last - 1))
            if err is not None:
                raise err
        self._warn_damage(first, last)
        h0 = self.header0
        nslot = len(self._thread_ids)
        nsets = last - first
        chunk = h0.nchan * (2 if self.complex_data else 1)
        flat = kernels.decode_frames(
            dev, nsets, h0.payload_nbytes, self._coder, self.bps, chunk=chunk,
            nslot=nslot, src=src[first * nslot:last * nslot].contiguous(),
            complex_data=self.complex_data, fill_value=self.fill_value, out=into,
            within=self._within)
        if self.complex_data:
            flat = torch.view_as_complex(flat.view(-1, 2))
        return flat.reshape((nsets * self.samples_per_frame,) + tuple(self._decode_shape))

    def _get_index(self, header):
        """Frame-set index relative to header0 (vdif/base.py:386-390)."""
        return int((header['seconds'] - self.header0['seconds'])
                   * self._frame_rate
                   + header['frame_nr'] - self.header0['frame_nr'])

    def _find_last_header(self):
        """Last header of header0's thread, searching backwards from the end
        of the file (vdif/base.py:492-517); HeaderNotFoundError if the last
        two frame sets' worth of bytes hold none.  A file whose last whole
        frame stands on the fixed stride is looked at through the table of
        its headers; otherwise -- bytes went missing somewhere -- the search
        is the reference's: byte by byte backwards over two frame sets, a
        candidate counting only with headers one frame before and after it."""
        hw = self.fh_raw._header_table(self.header0)
        nfull = len(self._image()) // self._frame_nbytes
        look = 2 * len(self._file_threads) + 1

        def is_header(words):
            return not any(((int(w) ^ p) & m) for w, p, m in zip(words, self._pattern, self._mask))
        last = min(nfull, len(hw)) - 1
        if last >= 0 and is_header(hw[last]):
            for k in range(last, max(-1, nfull - 1 - look), -1):
                words = hw[k]
                if ((int(words[3]) >> 16) & 0x3ff) != self.header0['thread_id']:
                    continue
                if not is_header(words):
                    continue
                return VDIFHeader(words, edv=self.header0.edv, verify=False)
        image = self._image()
        nw = self.header0.nbytes // 4
        with self.fh_raw.temporary_offset(max(0, len(image) - self._frame_nbytes)):
            found = self.fh_raw.locate_frames(self.header0, forward=False, maximum=2 * self._set_nbytes,
                                              check=(-1, 1))
        for p in found:
            words = np.frombuffer(np.asarray(image[p:p + 4 * nw]).tobytes(), '<u4')
            if ((int(words[3]) >> 16) & 0x3ff) == self.header0['thread_id']:
                return VDIFHeader(words, edv=self.header0.edv, verify=False)
        raise HeaderNotFoundError(
            "corrupt VDIF? No thread_id={0} frame in last {1} bytes."
            .format(self.header0['thread_id'], 2 * self._set_nbytes))

    def _squeeze_and_subset(self, data):
        # threads were already selected on read (vdif/base.py:519-528)
        if self._within_np is not None:            # ... and so were the channels, in the kernel
            return data.reshape(data.shape[:1] + self.sample_shape)
        if self.squeeze:
            data = data.reshape(data.shape[:1]
                                + tuple(sh for sh in data.shape[1:] if sh > 1))
        if self._frameset_subset:
            sub = tuple(torch.as_tensor(np.asarray(s), device=data.device)
                        if isinstance(s, (list, np.ndarray)) else s
                        for s in self._frameset_subset)
            data = data[(slice(None),) + sub]
        return data

    _window = None          # kernels.VDIFWindow: argument blocks of the fused window call
    _window_unverified = None   # ... with an empty invariant mask (verify=False)

    def _prepare_window_state(self, device):
        super()._prepare_window_state(device)
        if self._thread_slot is None:
            self._thread_slot = kernels.thread_slot_map(self._thread_ids, device)

    def _side_state_key(self):
        ts, w = self._thread_slot, self._within
        return (None if ts is None else ts.data_ptr(), None if w is None else w.data_ptr())

    def _process_window(self, dbuf, first_set, last_set, out_flat):
        """scan -> index -> verification -> decode for frame sets [first_set,
        last_set): one library call (bb_vdif_read_window)."""
        h0 = self.header0
        nsets = last_set - first_set
        nthread_file = len(self._file_threads)
        nframes = min(nsets * nthread_file, dbuf.numel() // self._frame_nbytes)
        if self._thread_slot is None:
            self._thread_slot = kernels.thread_slot_map(self._thread_ids,
                                                        dbuf.device)
        # verify=False: headers are not looked at beyond what places a frame (thread,
        # frame number) -- a damaged sync pattern does not make its frame fill, as in the
        # reference, whose verification is what is switched off (vdif/base.py:530-534):
        # a window whose invariant mask is empty
        which = '_window' if self.verify else '_window_unverified'
        w = getattr(self, which)
        if w is None:
            w = kernels.VDIFWindow(
                self._frame_nbytes, h0.nbytes, self._pattern,
                self._mask if self.verify else [0] * len(self._mask), h0['seconds'], self._frame_rate,
                h0.payload_nbytes, self._coder, self.bps, h0.nchan * (2 if self.complex_data else 1),
                len(self._thread_ids), self.complex_data, self.fill_value)
            setattr(self, which, w)
        if w.fill_value != self.fill_value:
            w.set_fill(self.fill_value)
        nbad = verified = None
        if self.verify:
            # (the verification is queued BEFORE the decode and an event recorded
            # behind it: read() waits for this verdict only -- `_resolve_checks`)
            nbad, verified = self._verdict_targets()
        w.run(dbuf, h0['frame_nr'] + first_set, nframes, self._thread_slot, nsets, self._within, out_flat,
              nthread_file, nframes, nbad, verified, scan_stream=self._scan_side)
        if self.verify:
            self._note_checked(nframes, missing=nsets * nthread_file - nframes)

    # -- frame index as a first-class object (multi-GPU sharding, parallel.py)
    def build_index(self, first=0, last=None):
        """Dense device index of frame sets [first, last): int64 payload
        offsets INTO THE FILE for every (frame set, selected thread), -1 where
        a frame is missing/invalid.  Streams the headers' windows through HBM
        and runs bb_vdif_scan + bb_build_index on each."""
        from ..staging import WindowPipeline
        h0 = self.header0
        nsets_total = self._nsample // self.samples_per_frame
        last = nsets_total if last is None else last
        nslot = len(self._thread_ids)
        nthread_file = len(self._file_threads)
        src = torch.full(((last - first) * nslot,), -1, dtype=torch.int64, device='cuda')
        if self._thread_slot is None:
            self._thread_slot = kernels.thread_slot_map(self._thread_ids, src.device)
        image = self._image()
        per_win = max(1, self.window_bytes // self._set_nbytes)
        pipe = WindowPipeline(image, per_win * self._set_nbytes)
        ranges, spans = [], []
        for s in range(first, last, per_win):
            e = min(last, s + per_win)
            ranges.append((s * self._set_nbytes, min(e * self._set_nbytes, len(image))))
            spans.append((s, e))

        def process(dbuf, i):
            s, e = spans[i]
            nframes = min((e - s) * nthread_file, dbuf.numel() // self._frame_nbytes)
            recs = kernels.vdif_scan(dbuf, nframes, self._frame_nbytes, h0.nbytes,
                                     self._pattern, self._mask, h0['seconds'],
                                     h0['frame_nr'] + s, self._frame_rate, set_nframes=nthread_file)
            part = kernels.build_index(recs, e - s, nslot, self._thread_slot)
            part = torch.where(part >= 0, part + ranges[i][0], part)
            src[(s - first) * nslot:(e - first) * nslot] = part

        pipe.run(ranges, process)
        pipe.drain()
        return src

    def decode_with_index(self, src, byte_lo, byte_hi, nsets):
        """Decode `nsets` frame sets given a dense index whose offsets are
        relative to file bytes [byte_lo, byte_hi) (see parallel.local_index)."""
        h0 = self.header0
        nslot = len(self._thread_ids)
        dbuf = kernels.to_device_bytes(np.asarray(self._image()[byte_lo:byte_hi]))
        if dbuf.numel() % 4:
            dbuf = torch.nn.functional.pad(dbuf, (0, 4 - dbuf.numel() % 4))
        chunk = h0.nchan * (2 if self.complex_data else 1)
        if dbuf.numel() == 0:
            dbuf = torch.zeros(4, dtype=torch.uint8, device='cuda')
        flat = kernels.decode_frames(
            dbuf, nsets, h0.payload_nbytes, self._coder, self.bps, chunk=chunk,
            nslot=nslot, src=src.to(dbuf.device), complex_data=self.complex_data,
            fill_value=self.fill_value)
        if self.complex_data:
            flat = torch.view_as_complex(flat.view(-1, 2))
        return flat.reshape((nsets * self.samples_per_frame, nslot, h0.nchan))


class VDIFStreamWriter(GPUStreamWriterBase):
    """VDIF stream writer (vdif/base.py:756-807): samples of shape
    ``(n, nthread, nchan)`` are packed on the GPU; one frame per thread and
    time step is written, threads in increasing thread_id order."""
    _sample_shape_fields = ('nthread', 'nchan')

    def __init__(self, fh_raw, header0=None, sample_rate=None, nthread=1,
                 squeeze=True, **kwargs):
        if header0 is None:
            kwargs.setdefault('edv', False)
            if sample_rate is not None and kwargs['edv'] in (1, 3):
                kwargs.setdefault('sample_rate', sample_rate)
            time = kwargs.pop('time', None)
            header0 = VDIFHeader.fromvalues(verify=False, **kwargs)
            if sample_rate is None:
                sample_rate = header0.sample_rate
            if time is not None and sample_rate is not None:
                if 'ref_epoch' not in kwargs and 'ref_time' not in kwargs:
                    header0.ref_time = time             # (a time without an epoch brings its own)
                header0.set_time(time, frame_rate=sample_rate / header0.samples_per_frame)
            header0.verify()
        elif kwargs:                    # (header keywords next to a header: the reference's TypeError)
            raise TypeError("__init__() got an unexpected keyword argument '{}'".format(sorted(kwargs)[0]))
        if sample_rate is None:
            sample_rate = header0.sample_rate
        if sample_rate is None:
            raise ValueError("the sample rate must be passed either "
                             "explicitly, or through the header if it "
                             "can be stored there.")
        super().__init__(fh_raw, header0, sample_rate=sample_rate,
                         samples_per_frame=header0.samples_per_frame,
                         unsliced_shape=(nthread, header0.nchan),
                         bps=header0.bps, complex_data=header0.complex_data,
                         squeeze=squeeze)
        self._frame_rate = int(round(self.sample_rate / self.samples_per_frame))
        self._start_time = header0.get_time(frame_rate=self._frame_rate)
        self._coder = (_lib.CODER_MARK5B if header0.edv == 0xab else _lib.CODER_VDIF)

    def _write_frames(self, data, valid):
        nthread, nchan = self._unsliced_shape
        spf = self.samples_per_frame
        nsets = data.shape[0] // spf
        # (set, sample, thread, chan) -> (set, thread, sample, chan): payload order
        block = data.reshape(nsets, spf, nthread, nchan).permute(0, 2, 1, 3).contiguous()
        packed = kernels.encode_flat(block, self._coder, self.bps)
        h = self.header0.copy()
        idx = self.header0['frame_nr'] + self._nframes_written
        h['seconds'] = self.header0['seconds'] + idx // self._frame_rate
        h['frame_nr'] = idx % self._frame_rate
        # thread ids are 0..n-1 whatever header0 holds (vdif/frame.py:277-285)
        words = frame_header_words(h, nsets, list(range(nthread)), self._frame_rate)
        heads = words.view(np.uint8).reshape(nsets, nthread, -1)
        heads[~np.asarray(valid, bool), :, 3] |= 0x80          # invalid_data bit of the whole set
        self._emit_frames(heads.reshape(nsets * nthread, -1), packed)


def _adopt_header(h):
    """The reference's VDIFHeader (anything with `words` that is not ours) ->
    ours, same words and EDV, mutable like the copy a writer works on."""
    if isinstance(h, VDIFHeader) or not hasattr(h, 'words'):
        return h
    new = VDIFHeader([int(w) for w in h.words], edv=getattr(h, 'edv', None), verify=False)
    return new.copy()


def _header_keywords():
    from .header import VDIF_HEADER_CLASSES
    names = {'edv', 'verify', 'time', 'frame_rate', 'header0', 'nthread', 'squeeze', 'file_size'}
    for cls in VDIF_HEADER_CLASSES.values():
        names.update(cls._properties)
        names.update(getattr(cls, '_header_parser', {}).keys())
    return {n.lower() for n in names}


open = FormatOpener('VDIF', {'rb': VDIFFileReader, 'wb': VDIFFileWriter, 'rs': VDIFStreamReader,
                             'ws': VDIFStreamWriter}, adopt_header=_adopt_header,
                    header_keywords=_header_keywords)
open.__doc__ = """Open VDIF file(s): ``'rb'`` gives a `VDIFFileReader`, ``'rs'`` a
`VDIFStreamReader`, ``'ws'`` a `VDIFStreamWriter` (vdif/base.py:810-884).
`name` may be a file name, a file handle, a list of names or a ``{file_nr}``
template; with ``'ws'`` and ``file_size=`` a sequence of files is written."""
