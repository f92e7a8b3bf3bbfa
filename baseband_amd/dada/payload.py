"""DADA payloads: int8 -> float32/complex64 on the GPU
(dada/payload.py:21-89).  Standard payloads are already stored
(time, pol, chan[, re/im]) and use the flat cast of ``bb_decode_frames``;
MeerKAT beamformer (MKBF) heaps go through ``bb_decode_i8_tiled``."""
import operator
from collections import namedtuple

import numpy as np
import torch

from .. import _lib, kernels
from ..base.payload import PayloadBase, RowSetMixin

__all__ = ['DADAPayload', 'MKBFPayload', 'decode_i8_rows']


def decode_i8_rows(dbuf, byte0, row_nbytes, start, stop):
    """Flat int8 -> float32 of rows [start, stop) (row = one complete sample)
    located at byte0 in dbuf; handles starts/ends that are not dword aligned
    by decoding the enclosing aligned range."""
    b0 = byte0 + start * row_nbytes
    b1 = byte0 + stop * row_nbytes
    if b1 == b0:
        return torch.empty(0, dtype=torch.float32, device=dbuf.device)
    lo = b0 - b0 % 4
    hi = -(-b1 // 4) * 4
    if hi > dbuf.numel():
        dbuf = torch.nn.functional.pad(dbuf, (0, hi - dbuf.numel()))
    flat = kernels.decode_frames(dbuf, 1, hi - lo, _lib.CODER_INT, 8, src0=lo)
    return flat[b0 - lo:b1 - lo]


class DADAPayload(RowSetMixin, PayloadBase):
    _memmap = True
    _dtype_word = np.dtype('<u4')
    _coder_id = _lib.CODER_INT
    _sample_shape_maker = namedtuple('SampleShape', 'npol, nchan')

    def __new__(cls, words, *, header=None, **kwargs):
        if header is not None and header.get("INSTRUMENT") == "MKBF":
            cls = MKBFPayload
        return super().__new__(cls)

    def __init__(self, words, *, header=None, sample_shape=(), bps=8,
                 complex_data=False):
        if words.dtype != self._dtype_word:
            words = np.asarray(words).view(np.uint8)
            self._dtype_word = words.dtype
        super().__init__(words, header=header, sample_shape=sample_shape,
                         bps=bps, complex_data=complex_data)

    def _rows(self, start, stop):
        if self.bps not in (8, 32):
            raise KeyError(self.bps)
        npol, nchan = self.sample_shape
        row = npol * nchan * (2 if self.complex_data else 1)
        if self.bps == 32:
            # EXTENSION: NBIT 32 = float32 passthrough (not in the reference)
            # (bb_copy_frames, csrc/k_copy.h)
            words = self._device_words()
            nb = (stop - start) * row * 4
            if nb and words.data_ptr() % 4 == 0:
                flat = kernels.copy_frames(words, 1, nb, src0=start * row * 4)
            else:
                flat = words[start * row * 4:stop * row * 4].clone().view(torch.float32)
        else:
            flat = decode_i8_rows(self._device_words(), 0, row, start, stop)
        if self.complex_data:
            flat = torch.view_as_complex(flat.reshape(-1, 2))
        return flat.reshape(stop - start, npol, nchan)

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
            data = self._rows(start, max(stop, start))[::step]
            if sample_index:
                data = data[(slice(None),) + sample_index]
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
            data = self._rows(first, first + 1)[0]
            if sample_index:
                data = data[sample_index]
        return data

    data = property(__getitem__, doc="Full decoded payload (device tensor).")

    def _store_rows(self, lo, hi, block):
        if self.bps != 8:
            raise ValueError(f"{type(self).__name__} cannot encode data with {self.bps} bits")
        comp = torch.view_as_real(block) if block.is_complex() else block
        enc = kernels.encode_flat(comp, _lib.CODER_INT, 8).cpu().numpy().view(np.int8)
        self.words.view(np.int8).reshape(len(self), -1)[lo:hi] = enc.reshape(hi - lo, -1)

    @classmethod
    def fromdata(cls, data, header=None, bps=8):
        """(nsample, npol, nchan) samples -> int8 words (dada/payload.py:17-18,
        80-89), rounded, clipped and packed by the GPU int8 encoder; MKBF
        headers get the (heap, pol, chan, 256, re/im) heap order."""
        data = kernels.as_device_samples(data)
        if (bps if header is None else header.bps) != 8:
            raise ValueError(f"{cls.__name__} cannot encode data with {bps} bits")
        comp = torch.view_as_real(data) if data.is_complex() else data
        if header is not None and header.get("INSTRUMENT") == "MKBF":
            npol, nchan = header.sample_shape
            comp = comp.reshape(-1, 256, npol, nchan, 2).movedim(1, 3)
        words = kernels.encode_flat(comp, _lib.CODER_INT, 8).cpu().numpy()
        if header is not None:
            return cls(words, header=header)
        return cls(words, sample_shape=tuple(data.shape[1:]), bps=bps,
                   complex_data=data.is_complex())


class MKBFPayload(DADAPayload):
    """Heaps of 256 samples stored (heap, pol, chan, 256, re/im)
    (dada/payload.py:54-89)."""

    _row_granule = 256

    def _store_rows(self, lo, hi, block):
        npol, nchan = self.sample_shape
        comp = torch.view_as_real(block).reshape(-1, 256, npol, nchan, 2).movedim(1, 3)
        enc = kernels.encode_flat(comp, _lib.CODER_INT, 8).cpu().numpy().view(np.int8)
        self.words.view(np.int8).reshape(len(self) // 256, -1)[lo // 256:hi // 256] = (
            enc.reshape((hi - lo) // 256, -1))

    def _rows(self, start, stop):
        if self.bps != 8 or not self.complex_data:
            raise KeyError(self.bps)
        npol, nchan = self.sample_shape
        flat = kernels.decode_i8_tiled(self._device_words(), 1, _lib.LAYOUT_MKBF,
                                       npol, nchan, len(self), start, stop)
        return torch.view_as_complex(flat.view(-1, 2)).reshape(stop - start, npol, nchan)
