"""Mark 5B payloads: GPU decode of 1/2-bit sign/magnitude samples
(mark5b/payload.py:27-94,112-150).  Fixed 10000-byte payloads."""
from collections import namedtuple

import numpy as np

from .. import _lib
from ..base.payload import PayloadBase
from ..base import encoding as enc

__all__ = ['Mark5BPayload', 'encode_mark5b']


def encode_mark5b(comp, bps):
    """float32 components -> packed bytes.  2-bit codes are re-ordered so
    that the sign sits on the even and the magnitude on the odd bit stream
    (mark5b/payload.py:97-106); 1 bit stores the sign bit."""
    if bps == 1:
        return enc.pack_codes(np.signbit(np.asarray(comp)).astype(np.uint8), 1)
    if bps == 2:
        reorder = np.array([0, 2, 1, 3], dtype=np.uint8)
        return enc.pack_codes(reorder[enc.codes_2bit(comp)], 2)
    raise ValueError(f"Mark5BPayload cannot encode data with {bps} bits")


class Mark5BPayload(PayloadBase):
    _nbytes = 10000
    _coder_id = _lib.CODER_MARK5B
    _sample_shape_maker = namedtuple('SampleShape', 'nchan')

    def __init__(self, words, header=None, *, sample_shape=(1,), bps=2,
                 complex_data=False):
        if complex_data:
            raise ValueError("Mark5B format does not support complex data.")
        super().__init__(words, sample_shape=sample_shape, bps=bps,
                         complex_data=False)

    def _decode(self, byte_start, byte_stop):
        if self.bps not in (1, 2):
            raise KeyError(self.bps)
        return super()._decode(byte_start, byte_stop)

    @classmethod
    def _encode_data(cls, data, bps, **kwargs):
        return encode_mark5b(enc.components(data), bps).view('<u4')

    @classmethod
    def fromdata(cls, data, header=None, bps=2):
        if data.dtype.kind == 'c':
            raise ValueError("Mark5B format does not support complex data.")
        words = cls._encode_data(np.asarray(data), bps)
        return cls(words, sample_shape=data.shape[1:], bps=bps)
