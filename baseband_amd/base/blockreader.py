"""Stream reader for byte-aligned block formats (GUPPI, DADA).

These formats have few, large frames (tens of MiB) with ASCII headers, so
there is no device-side header scan: the host walks the headers and the read
is cut into pieces ``(frame, first row, last row)`` exactly as the reference's
loop would take them (base/base.py:957-967 with the per-format ``_get_frame``
rules).  Runs of frames with the same row range are staged frame-by-frame
through the pinned pipeline and decoded by one kernel launch per window.
"""
import numpy as np
import torch

from .. import kernels
from ..placement import empty_output
from ..staging import WindowPipeline
from .base import GPUStreamReaderBase, _to_host_array


class BlockStreamReader(GPUStreamReaderBase):
    # subclasses set: _frame_nbytes, _header_nbytes, _file_offset0, _nframes
    def _pieces(self, offset, count):
        """[(frame, row_lo, row_hi), ...] covering `count` samples from
        sample `offset`."""
        raise NotImplementedError

    def _decode_window(self, dbuf, nframes, row_lo, row_hi, out_flat,
                       payload_offset, frame_stride, first_frame):
        raise NotImplementedError

    def _row_range_source(self, frame, a, b):
        """For rows [a, b) of `frame`: ``(pieces, decode)`` where `pieces` is a
        list of (byte offset in the file image, nbytes) that together hold
        those rows and ``decode(dbuf, out_flat)`` decodes them from a device
        buffer in which the pieces lie back to back -- or None when the format
        has no such shortcut (the whole frame is staged then)."""
        return None

    _touched = None         # sample offset at which the previous small request ended
    _prefetch = None        # (frame, future) of a frame on its way to HBM in the background
    prefetch_next = True    # set False to stage every frame when it is first needed

    def _start_prefetch(self, frame):
        """A sequential loop of small reads is inside frame - 1: bring `frame`
        to HBM on the worker thread meanwhile (`staging.upload_in_background`)."""
        if (not self.prefetch_next or frame >= self._nframes or self._resident_bytes() is not None
                or (self._prefetch is not None and self._prefetch[0] == frame)):
            return
        from ..staging import upload_in_background
        image = self._image()
        lo, nbytes = self._frame_span(frame)
        self._prefetch = (frame, upload_in_background(image, lo, min(lo + nbytes, len(image))))

    def _take_prefetch(self, frame):
        """The device bytes of `frame` if they were prefetched, else None."""
        pending, self._prefetch = self._prefetch, None
        if pending is None:
            return None
        dev, done = pending[1].result()             # (also waits out a prefetch nobody wants)
        if pending[0] != frame:
            return None
        torch.cuda.current_stream().wait_event(done)
        return dev

    def close(self):
        if self._prefetch is not None:
            try:
                self._prefetch[1].result()
            except Exception:
                pass
            self._prefetch = None
        super().close()
    _chan_lo = 0            # first channel decoded (`_plan_channel_range`)
    _sel = None             # (first pol, pols kept, int32 channel list): a selection that is not a plain range
    _cmap_dev = None

    def _plan_channel_range(self):
        """For (pol, chan) samples decoded by bb_decode_i8_tiled: fold a
        `subset` into the decode instead of decoding whole blocks and indexing
        afterwards, as the reference does (base/base.py:706-717 after
        guppi/payload.py:90-102, dada/payload.py:76-79).

        * all polarisations and a contiguous RANGE of channels: the same decode
          with the payload entered at the first kept channel and fewer channels
          (`nchan_stored` tells the kernel the strides); the other channels are
          not written and -- where the format stores channels apart (GUPPI
          channels-first) -- not read either.  Every kernel takes this.
        * any channel LIST (gaps, any order) and / or one of two polarisations:
          `chan_map` / `pol_first` of bb_tiled_params (`_sel`); taken by the
          fast transposing kernel only -- `_tiled_decode` falls back to decoding
          whole blocks and indexing when the library answers ENOTSUP.

        Checked by value, like `_plan_channel_select`: the subset is applied to
        arrays of polarisation and channel numbers."""
        if not self.subset or len(self._unsliced_shape) != 2:
            return
        npol, nchan = self._unsliced_shape
        chan = np.broadcast_to(np.arange(nchan), (npol, nchan))
        pol = np.broadcast_to(np.arange(npol)[:, np.newaxis], (npol, nchan))

        def view(a):
            a = np.ascontiguousarray(a)[np.newaxis]
            if self.squeeze:
                a = a.reshape(a.shape[:1] + tuple(s for s in a.shape[1:] if s > 1))
            return a[(slice(None),) + tuple(self.subset)]
        try:
            c, p = view(chan).reshape(-1), view(pol).reshape(-1)
        except Exception:
            return
        if c.size == 0:
            return
        # the polarisations kept: a run pf, pf + 1, ... each with the same channel list
        npk = int(np.unique(p).size)
        if c.size % npk:
            return
        m, pf, lo = c.size // npk, int(p[0]), int(c[0])
        cl = c[:m]
        if not (np.array_equal(p, np.repeat(np.arange(pf, pf + npk), m))
                and np.array_equal(c, np.tile(cl, npk))):
            return
        whole_pols = npk == npol
        if whole_pols and np.array_equal(cl, np.arange(lo, lo + m)):
            if m == nchan:
                return                              # everything, in order: nothing to fold
            self._chan_lo = lo
            self._decode_shape = (npol, m)
            self._within_np = np.arange(lo, lo + m, dtype=np.int32)     # (the decode applies the subset)
            return
        if not (whole_pols or (npol == 2 and npk == 1)) or m % 2 or m > 65536:
            return                                  # (the kernel writes channel pairs; odd counts index afterwards)
        self._sel = (pf, npk, cl.astype(np.int32))
        self._decode_shape = (npk, m)
        self._within_np = cl.astype(np.int32)

    def _tiled_decode(self, dbuf, nframes, layout, ntime, a, b, src0, src_stride, out_flat):
        """Times [a, b) of `nframes` blocks of `ntime` stored times each (payloads
        at ``src0 + i * src_stride``) -> `out_flat`, honouring the planned channel
        range or selection."""
        npol_s, nchan_s = self._unsliced_shape
        if self._sel is None:
            skip = kernels.tiled_channel_skip(layout, npol_s, ntime, self._chan_lo)
            kernels.decode_i8_tiled(dbuf, nframes, layout, npol_s, self._decode_shape[-1], ntime, a, b,
                                    src0=src0 + skip, src_stride=src_stride, out=out_flat,
                                    nchan_stored=nchan_s)
            return
        pf, npk, cl = self._sel
        if self._cmap_dev is None or self._cmap_dev.device != dbuf.device:
            self._cmap_dev = torch.from_numpy(cl).to(dbuf.device)
        try:
            kernels.decode_i8_tiled(dbuf, nframes, layout, npk, cl.size, ntime, a, b, src0=src0,
                                    src_stride=src_stride, out=out_flat, nchan_stored=nchan_s,
                                    npol_stored=npol_s, pol_first=pf, chan_map=self._cmap_dev)
            return
        except KeyError:
            pass
        # a geometry the selecting kernel does not take (unaligned payloads):
        # whole blocks, then the index -- the reference's order -- in pieces of
        # at most 1 GiB of decoded samples
        rows = b - a
        per = max(1, (1 << 28) // max(1, rows * npol_s * nchan_s * 2))
        idx = self._cmap_dev.long()
        n_out = rows * npk * cl.size * 2
        for f0 in range(0, nframes, per):
            n = min(per, nframes - f0)
            tmp = kernels.decode_i8_tiled(dbuf, n, layout, npol_s, nchan_s, ntime, a, b,
                                          src0=src0 + f0 * src_stride, src_stride=src_stride)
            part = tmp.view(n * rows, npol_s, nchan_s, 2)[:, pf:pf + npk][:, :, idx]
            out_flat[f0 * n_out:(f0 + n) * n_out] = part.reshape(-1)

    def _frame_span(self, frame):
        """(byte offset of the frame in the file, number of bytes to stage)."""
        return (self._file_offset0 + frame * self._frame_nbytes,
                self._frame_nbytes)

    read_through = True     # queue a read-only pass over a request's frames in front of their decode (kernels.touch)

    def read(self, count=None, out=None):
        if self.closed:
            raise ValueError("I/O operation on closed stream.")
        samples_left = self.shape[0] - self.offset
        if out is None:
            if count is None or count < 0:
                count = max(0, samples_left)
        else:
            assert tuple(out.shape[1:]) == self.sample_shape, (
                "'out' must have trailing shape {}".format(self.sample_shape))
            count = out.shape[0]
        if count > samples_left:
            raise EOFError("cannot read from beyond end of input.")
        kernels.require_gpu()
        ncomp = 2 if self.complex_data else 1
        row = int(np.prod(self._decode_shape)) * ncomp
        # pieces land exactly where they belong, so a suitable `out` tensor is
        # decoded into directly
        direct = (isinstance(out, torch.Tensor) and out.is_cuda and out.is_contiguous()
                  and count > 0 and (not self.subset or self._within_np is not None)
                  and out.dtype == (torch.complex64 if self.complex_data else torch.float32))
        pieces = self._pieces(self.offset, count)
        image = self._image()
        resident = self._resident_bytes()
        if resident is not None and pieces and self.read_through:
            # The frames of this request are in HBM: a pass that only reads them is queued NOW,
            # before the output is drawn and the launches are prepared; the decode behind it finds
            # its input -- a fifth of an int8 decode's traffic -- in the memory-side cache and runs
            # 20-25 % faster (profiles/r06dm_exp_cached_input_int8.log)
            lo = self._frame_span(pieces[0][0])[0]
            last_lo, last_n = self._frame_span(pieces[-1][0])
            hi = min(last_lo + last_n, resident.numel())
            if 2 * count * self._frame_nbytes >= (hi - lo) * self.samples_per_frame:    # (most of it is wanted)
                kernels.touch(resident, lo, hi - lo)
        if direct:
            flat = (torch.view_as_real(out) if self.complex_data else out).reshape(-1)
        else:
            flat = empty_output(count * row, torch.float32)
        # merge consecutive frames that use the same row range into runs
        runs = []
        for f, a, b in pieces:
            if runs and runs[-1][1] == f and tuple(runs[-1][2:]) == (a, b):
                runs[-1][1] = f + 1
            else:
                runs.append([f, f + 1, a, b])
        done = 0
        # requests of less than one frame (loops of small reads): the frame is
        # staged once and kept in HBM; following requests inside it only launch
        # kernels instead of staging the whole block again for every call
        if count < self.samples_per_frame and len(runs) <= 2 and all(r[1] == r[0] + 1 for r in runs):
            from ..staging import upload
            for f0, f1, a, b in runs:
                o = flat[done * row:(done + (b - a)) * row]
                done += b - a
                cached = self._ahead is not None and self._ahead[0] == f0
                # random access: stage just the rows asked for; a request that
                # continues where the previous small one ended (a sequential
                # loop) brings the whole frame in
                source = None if cached or self._touched == self.offset else self._row_range_source(f0, a, b)
                if resident is not None:
                    # the frame is in HBM already: decode from where it lies
                    lo, nbytes = self._frame_span(f0)
                    self._decode_window(self._device_window(resident, lo, min(lo + nbytes, resident.numel())),
                                        1, a, b, o, self._header_nbytes, self._frame_nbytes, f0)
                    continue
                if source is not None:
                    pieces, decode = source
                    host = np.concatenate([image[lo:lo + n] for lo, n in pieces]) if len(pieces) > 1 \
                        else image[pieces[0][0]:pieces[0][0] + pieces[0][1]]
                    decode(upload(host), o)
                    continue
                if not cached:
                    dev = self._take_prefetch(f0)
                    if dev is None:
                        lo, nbytes = self._frame_span(f0)
                        dev = upload(image[lo:min(lo + nbytes, len(image))])
                    self._ahead = (f0, dev)
                    # a sequential loop: the next frame travels while this one is consumed
                    self._start_prefetch(f0 + 1)
                self._decode_window(self._ahead[1], 1, a, b, o, self._header_nbytes,
                                    self._frame_nbytes, f0)
            runs = []
            self._touched = self.offset + count
        for f0, f1, a, b in runs:
            if resident is not None:
                # ONE launch for the whole run of frames, straight from HBM
                lo = self._frame_span(f0)[0]
                last_lo, last_n = self._frame_span(f1 - 1)
                hi = min(last_lo + last_n, resident.numel())
                o = flat[done * row:(done + (f1 - f0) * (b - a)) * row]
                self._decode_window(self._device_window(resident, lo, hi), f1 - f0, a, b, o,
                                    self._header_nbytes, self._frame_nbytes, f0)
                done += (f1 - f0) * (b - a)
                continue
            off0, nbytes = self._frame_span(f0)
            per_win = max(1, self.window_bytes // self._frame_nbytes)
            cap = max(per_win * self._frame_nbytes, nbytes)
            if self._pipeline is None or self._pipeline.cap < cap:
                self._pipeline = WindowPipeline(image, cap)
            ranges, spans = [], []
            for s in range(f0, f1, per_win):
                e = min(f1, s + per_win)
                lo = self._frame_span(s)[0]
                last_lo, last_n = self._frame_span(e - 1)
                ranges.append((lo, min(last_lo + last_n, len(image))))
                spans.append((s, e))
            base = done

            def process(dbuf, i, base=base, a=a, b=b, f0=f0):
                s, e = spans[i]
                o = flat[(base + (s - f0) * (b - a)) * row:
                         (base + (e - f0) * (b - a)) * row]
                self._decode_window(dbuf, e - s, a, b, o, self._header_nbytes,
                                    self._frame_nbytes, s)

            self._pipeline.run(ranges, process)
            done += (f1 - f0) * (b - a)
        assert done == count
        if direct:
            self.offset += count
            return out
        if self.complex_data:
            flat = torch.view_as_complex(flat.view(-1, 2))
        data = flat.reshape((count,) + tuple(self._decode_shape))
        data = self._squeeze_and_subset(data)
        self.offset += count
        if out is None:
            if self.host_results:           # (what the plugin modules switch on: NumPy arrays, base/base.py)
                from ..staging import download_new
                return download_new(data)
            return data
        if isinstance(out, torch.Tensor):
            out.copy_(data)
        else:
            _to_host_array(data, out)
        return out
