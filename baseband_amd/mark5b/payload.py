"""Mark 5B payloads: GPU decode of 1/2-bit sign/magnitude samples
(mark5b/payload.py:27-94,112-150).  Fixed 10000-byte payloads."""
from collections import namedtuple


from .. import _lib
from ..base.payload import PayloadBase

__all__ = ['Mark5BPayload']


class Mark5BPayload(PayloadBase):
    complex_data = False                # the format has real samples only
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
    def fromdata(cls, data, header=None, bps=2):
        """Pack (nsample, nchan) samples on the GPU (Mark 5B coder)."""
        from .. import kernels
        data = kernels.as_device_samples(data)
        if data.is_complex():
            raise ValueError("Mark5B format does not support complex data.")
        words = cls._encode_device(data, bps)
        return cls(words, sample_shape=tuple(data.shape[1:]), bps=bps)
