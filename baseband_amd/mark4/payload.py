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
from ..base import encoding as enc
from .header import MARK4_DTYPES
from ._bitmaps import BITMAPS

__all__ = ['Mark4Payload']


class Mark4Payload(PayloadBase):
    _dtype_word = None
    _sample_shape_maker = namedtuple('SampleShape', 'nchan')

    def __init__(self, words, header=None, *, sample_shape=(1,), bps=2,
                 fanout=1, magnitude_bit=None, complex_data=False):
        if header is not None:
            magnitude_bit = header.magnitude_signature()
            bps = header.bps
            ntrack = header.ntrack
            fanout = header.fanout
            sample_shape = (ntrack // (bps * fanout),)
            self._nbytes = header.payload_nbytes
        else:
            ntrack = sample_shape[0] * bps * fanout
            magnitude_bit = None
        if complex_data:
            raise ValueError("Mark4 format does not support complex data.")
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

    @classmethod
    def fromdata(cls, data, header):
        """Encode (nsample, nchan) data with the header's track layout."""
        on_gpu = isinstance(data, torch.Tensor) and data.is_cuda
        if isinstance(data, torch.Tensor) and not on_gpu:
            data = data.numpy()
        if (data.is_complex() if on_gpu else data.dtype.kind == 'c'):
            raise ValueError("Mark4 format does not support complex data.")
        if tuple(header.sample_shape) != tuple(data.shape[1:]):
            raise ValueError("header is for {0} channels but data has {1}"
                             .format(header.nchan, data.shape[-1]))
        if on_gpu:
            key = (header.nchan, header.magnitude_signature() or header.bps, header.fanout)
            maps = BITMAPS[key]
            words = kernels.encode_mark4(data, header.ntrack, maps['sign_bit'],
                                         maps['mag_bit']).cpu().numpy().view(header.stream_dtype)
        else:
            words = encode_mark4(data, header)
        return cls(words, header)


def encode_mark4(data, header):
    """float data (nsample, nchan) -> stream words, inverse of the bit maps:
    2-bit code = 2*sign + magnitude with levels {-Hi,-1,+1,+Hi}."""
    key = (header.nchan, header.magnitude_signature() or header.bps, header.fanout)
    maps = BITMAPS[key]
    ntrack = maps['ntrack']
    dtype = np.dtype(MARK4_DTYPES[ntrack])
    codes = enc.codes_2bit(np.asarray(data, dtype=np.float32))
    opw = ntrack // 2
    codes = codes.reshape(-1, opw).astype(np.uint64)
    sign = (codes >> np.uint64(1)) & np.uint64(1)
    mag = codes & np.uint64(1)
    words = np.zeros(codes.shape[0], dtype=np.uint64)
    for j in range(opw):
        words |= sign[:, j] << np.uint64(maps['sign_bit'][j])
        words |= mag[:, j] << np.uint64(maps['mag_bit'][j])
    return words.astype(dtype)
