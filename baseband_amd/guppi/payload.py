"""GUPPI payloads: int8 -> float32/complex64 on the GPU with the
(chan, time, pol) or (time, chan, pol) -> (time, pol, chan) permutation done
by ``bb_decode_i8_tiled`` (guppi/payload.py:20-110)."""
from collections import namedtuple

import numpy as np
import torch

from .. import _lib, kernels
from ..base.payload import PayloadBase, RowSetMixin

__all__ = ['GUPPIPayload']


class GUPPIPayload(RowSetMixin, PayloadBase):
    _dtype_word = np.dtype('int8')
    _memmap = True
    _coder_id = _lib.CODER_INT
    _sample_shape_maker = namedtuple('SampleShape', 'npol, nchan')

    def __init__(self, words, *, header=None, sample_shape=(), bps=8, complex_data=False,
                 channels_first=True):
        # storage order: the header's word when there is one
        if header is not None:
            channels_first = header.channels_first
        self.channels_first = channels_first
        super().__init__(words, header=header, sample_shape=sample_shape, bps=bps,
                         complex_data=complex_data)

    @classmethod
    def fromdata(cls, data, header=None, bps=8, channels_first=True):
        """(nsample, npol, nchan) samples -> int8 words in the on-disk order
        (guppi/payload.py:64-84): rounded, clipped and packed by the GPU int8
        encoder after the storage-order permutation."""
        data = kernels.as_device_samples(data)
        if header is not None:
            bps, channels_first = header.bps, header.channels_first
        if bps != 8:
            raise ValueError(f"{cls.__name__} cannot encode data with {bps} bits")
        comp = torch.view_as_real(data) if data.is_complex() else data.unsqueeze(-1)
        # (time, pol, chan, comp) -> (chan, time, pol, comp) or (time, chan, pol, comp)
        comp = comp.permute(2, 0, 1, 3) if channels_first else comp.permute(0, 2, 1, 3)
        words = kernels.encode_flat(comp, _lib.CODER_INT, 8).cpu().numpy().view(np.int8)
        if header is not None:
            return cls(words, header=header)
        return cls(words, sample_shape=tuple(data.shape[1:]), bps=bps,
                   complex_data=data.is_complex(), channels_first=channels_first)

    def _decode_rows(self, start, stop):
        """Rows [start, stop) -> (n, npol, nchan) device tensor."""
        if self.bps != 8:
            raise KeyError(self.bps)
        npol, nchan = self.sample_shape
        n = stop - start
        dbuf = self._device_words()
        if not self.complex_data:
            # real data: nchan == 1 by construction (OBSNCHAN == 1), so both
            # storage orders are (time, pol): flat cast
            if nchan != 1:
                raise KeyError("real-valued multi-channel GUPPI data")
            b0, b1 = start * npol, stop * npol
            lo = b0 - b0 % 4
            hi = min(-(-b1 // 4) * 4, dbuf.numel() - dbuf.numel() % 4)
            if hi < b1:                     # unaligned tail: pad a copy
                dbuf = torch.nn.functional.pad(dbuf, (0, 4))
                hi = -(-b1 // 4) * 4
            flat = kernels.decode_frames(dbuf, 1, hi - lo, _lib.CODER_INT, 8, src0=lo)
            return flat[b0 - lo:b1 - lo].reshape(n, npol, 1)
        layout = _lib.LAYOUT_GUPPI_CF if self.channels_first else _lib.LAYOUT_GUPPI_TF
        flat = kernels.decode_i8_tiled(dbuf, 1, layout, npol, nchan, len(self),
                                       start, stop)
        return torch.view_as_complex(flat.view(-1, 2)).reshape(n, npol, nchan)

    def __getitem__(self, item=()):
        if isinstance(item, tuple):
            sample_index = item[1:]
            first = item[0] if item else slice(None)
        else:
            sample_index, first = (), item
        nsample = len(self)
        if isinstance(first, slice):
            start, stop, step = first.indices(nsample)
            assert step > 0, "cannot deal with negative steps yet."
            data = self._decode_rows(start, max(stop, start))[::step]
        else:
            import operator
            try:
                first = operator.index(first)
            except Exception:
                raise TypeError("{0} object can only be indexed or sliced."
                                .format(type(self)))
            if first < 0:
                first += nsample
            if not (0 <= first < nsample):
                raise IndexError("{0} index out of range.".format(type(self)))
            data = self._decode_rows(first, first + 1)[0]
        if sample_index:
            data = data[(Ellipsis,) + sample_index] if not isinstance(first, slice) \
                else data[(slice(None),) + sample_index]
        return data

    def _store_rows(self, lo, hi, block):
        """Pack rows [lo, hi) given as (n, npol, nchan) and put them at their
        place in the on-disk order."""
        npol, nchan = self.sample_shape
        comp = torch.view_as_real(block) if block.is_complex() else block.unsqueeze(-1)
        ncomp = comp.shape[-1]
        comp = comp.permute(2, 0, 1, 3) if self.channels_first else comp.permute(0, 2, 1, 3)
        enc = kernels.encode_flat(comp, _lib.CODER_INT, 8).cpu().numpy().view(np.int8)
        words = self.words.view(np.int8)
        if self.channels_first:
            words.reshape(nchan, len(self), npol * ncomp)[:, lo:hi] = enc.reshape(
                nchan, hi - lo, npol * ncomp)
        else:
            words.reshape(len(self), -1)[lo:hi] = enc.reshape(hi - lo, -1)

    data = property(__getitem__, doc="Full decoded payload (device tensor).")
