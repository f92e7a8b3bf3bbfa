"""Mark 4 payloads: GPU track demultiplexing.

Mirror of ``Mark4Payload`` (mark4/payload.py:303-410).  The reference picks a
decoder function by ``(nchan, bps or magnitude-bit signature, fanout)``; here
the same key selects the sign/magnitude bit maps (`_bitmaps.BITMAPS`) handed
to ``bb_decode_mark4``.  An unknown key raises KeyError like the reference's
dict lookup.
"""
from collections import namedtuple

import numpy as np
import torch

from .. import kernels
from ..base.payload import PayloadBase
from .header import MARK4_DTYPES
from ._bitmaps import BITMAPS

__all__ = ['Mark4Payload']


class Mark4Payload(PayloadBase):
    complex_data = False                # the format has real samples only
    _dtype_word = None
    _sample_shape_maker = namedtuple('SampleShape', 'nchan')

    def __init__(self, words, header=None, *, sample_shape=(1,), bps=2,
                 fanout=1, magnitude_bit=None, complex_data=False):
        if complex_data:
            raise ValueError("Mark 4 samples are real: complex_data cannot be set.")
        if header is None:
            # every channel takes bps * fanout tracks
            ntrack, magnitude_bit = sample_shape[0] * bps * fanout, None
        else:
            ntrack, bps, fanout = header.ntrack, header.bps, header.fanout
            magnitude_bit = header.magnitude_signature()
            sample_shape = (ntrack // (bps * fanout),)
            self._nbytes = header.payload_nbytes
        self._dtype_word = np.dtype(MARK4_DTYPES[ntrack])
        self.fanout = fanout
        self.ntrack = ntrack
        super().__init__(words, sample_shape=sample_shape, bps=bps,
                         complex_data=False)
        self._coder = (self.sample_shape.nchan,
                       (self.bps if magnitude_bit is None else magnitude_bit),
                       self.fanout)

    @classmethod
    def fromfile(cls, fh, header=None, **kwargs):
        if header is not None:
            kwargs.setdefault('dtype', header.stream_dtype)
        return super().fromfile(fh, header=header, **kwargs)

    def _decode(self, byte_start, byte_stop):
        maps = BITMAPS[self._coder]          # KeyError for unsupported modes
        isz = self.words.itemsize
        nwords = (byte_stop - byte_start) // isz
        if nwords == 0:
            return torch.empty(0, dtype=torch.float32, device='cuda')
        return kernels.decode_mark4(
            self._device_words(), 1, self.ntrack, nwords, maps['sign_bit'],
            maps['mag_bit'], src0=byte_start)

    def _encode(self, data):
        maps = BITMAPS[self._coder]
        words = kernels.encode_mark4(data, self.ntrack, maps['sign_bit'], maps['mag_bit'])
        return words.cpu().numpy().view(self._dtype_word)

    @classmethod
    def fromdata(cls, data, header):
        """Encode (nsample, nchan) data with the header's track layout on the
        GPU (bb_encode_mark4; mark4/payload.py:138-300)."""
        data = kernels.as_device_samples(data)
        if data.is_complex():
            raise ValueError("Mark4 format does not support complex data.")
        if tuple(header.sample_shape) != tuple(data.shape[1:]):
            raise ValueError("header is for {0} channels but data has {1}"
                             .format(header.nchan, data.shape[-1]))
        key = (header.nchan, header.magnitude_signature() or header.bps, header.fanout)
        maps = BITMAPS[key]
        words = kernels.encode_mark4(data, header.ntrack, maps['sign_bit'], maps['mag_bit'])
        return cls(words.cpu().numpy().view(header.stream_dtype), header)
