"""Mark 4 frames: the header overwrites the first 160 stream words, so the
first ``160 * fanout`` samples of every frame read as ``fill_value``; a frame
with any error flag set is invalid (mark4/frame.py:23-263)."""
import operator

import numpy as np
import torch

from ..base.frame import FrameBase
from .header import Mark4Header
from .payload import Mark4Payload

__all__ = ['Mark4Frame']


class Mark4Frame(FrameBase):
    _header_class = Mark4Header
    _payload_class = Mark4Payload

    def __init__(self, header, payload, valid=None, verify=True):
        self.header, self.payload = header, payload
        # validity lives in the header's error flags: only touch them on request
        if valid is not None:
            self.valid = valid
        if verify:
            self.verify()

    @property
    def valid(self):
        h = self.header
        return not np.any(h['time_sync_error'] | h['internal_clock_error']
                          | h['processor_time_out_error']
                          | h['communication_error'])

    @valid.setter
    def valid(self, valid):
        if not self.header.mutable:
            self.header = self.header.copy()
        n = self.header.ntrack
        if valid:
            for key in ('time_sync_error', 'internal_clock_error',
                        'processor_time_out_error', 'communication_error'):
                self.header[key] = np.zeros(n, bool)
        else:
            self.header['communication_error'] = np.ones(n, bool)

    @classmethod
    def fromfile(cls, fh, ntrack, decade=None, ref_time=None, verify=True):
        header = Mark4Header.fromfile(fh, ntrack, decade=decade,
                                      ref_time=ref_time, verify=verify)
        payload = Mark4Payload.fromfile(fh, header=header)
        return cls(header, payload, verify=verify)

    @classmethod
    def fromdata(cls, data, header=None, verify=True, **kwargs):
        if header is None:              # (header from the keywords: mark4/frame.py:124-146 in the reference)
            header = Mark4Header.fromvalues(verify=verify, **kwargs)
        assert data.shape[0] == header.samples_per_frame
        start = header.nbytes * 8 // (header.ntrack // header.fanout)
        payload = Mark4Payload.fromdata(data[start:], header=header)
        return cls(header, payload, verify=verify)

    def __len__(self):
        """Samples including those overwritten by the header."""
        return self.header.samples_per_frame

    def __getitem__(self, item=()):
        if isinstance(item, str):
            return self.header.__getitem__(item)
        nsample = len(self)
        nfill = nsample - len(self.payload)
        if isinstance(item, tuple):
            sample_index = item[1:]
            first = item[0] if item else slice(None)
        else:
            sample_index, first = (), item
        if isinstance(first, slice):
            start, stop, step = first.indices(nsample)
            assert step > 0, "cannot deal with negative steps yet."
            picks = range(start, stop, step)
            npick = len(picks)
            ninvalid = len(range(start, min(stop, nfill), step)) if start < nfill else 0
            single = False
        else:
            try:
                first = operator.index(first)
            except Exception:
                raise TypeError("{0} object can only be indexed or sliced."
                                .format(type(self)))
            if first < 0:
                first += nsample
            if not (0 <= first < nsample):
                raise IndexError("{0} index out of range.".format(type(self)))
            start, step, npick = first, 1, 1
            ninvalid = 1 if first < nfill else 0
            single = True
        shape = (npick,) + tuple(self.sample_shape)
        if not self.valid or ninvalid == npick:
            data = self._fill(shape)
        else:
            pstart = start + ninvalid * step - nfill
            pstop = start + npick * step - nfill
            good = self.payload[pstart:pstop:step] if npick - ninvalid > 1 or not single \
                else self.payload[pstart:pstart + 1]
            if ninvalid:
                data = torch.cat([self._fill((ninvalid,) + tuple(self.sample_shape)), good])
            else:
                data = good
        if single:
            data = data[0]
        if sample_index:
            data = data[(Ellipsis,) + sample_index]
        return data

    def __setitem__(self, item, value):
        """Header key -> header; samples -> payload, ignoring whatever falls
        in the part overwritten by the header (mark4/frame.py:265-295).  The
        frame is updated as a whole on the GPU and its payload packed again."""
        if isinstance(item, str):
            return self.header.__setitem__(item, value)
        nfill = len(self) - len(self.payload)
        if not isinstance(value, torch.Tensor):
            value = torch.from_numpy(np.ascontiguousarray(value))
        assert value.ndim <= 2
        full = torch.cat([self._fill((nfill,) + tuple(self.sample_shape)),
                          self.payload.data])
        full[item] = value.to(device=full.device, dtype=full.dtype)
        self.payload[:] = full[nfill:]

    data = property(__getitem__, doc="Full decoded frame (device tensor).")
