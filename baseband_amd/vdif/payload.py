"""VDIF payloads: GPU decode of 1/2/4/8-bit offset-binary samples.

Mirror of the reference's ``VDIFPayload`` (vdif/payload.py:117-198).  The
per-bps decoder dict of the reference becomes the (coder, bps) pair handed to
``bb_decode_frames``; EDV 0xab payloads switch to the Mark 5B coder as in
vdif/payload.py:151-154.
"""
import numpy as np
from collections import namedtuple


from .. import _lib
from ..base.payload import PayloadBase

__all__ = ['VDIFPayload']


class VDIFPayload(PayloadBase):
    _coder_id = _lib.CODER_VDIF
    _sample_shape_maker = namedtuple('SampleShape', 'nchan')

    def __init__(self, words, header=None, sample_shape=(1,), bps=2,
                 complex_data=False):
        if header is not None and header.edv == 0xab:       # Mark5B payload
            self._coder_id = _lib.CODER_MARK5B
        super().__init__(words, header=header, sample_shape=sample_shape,
                         bps=bps, complex_data=complex_data)
        # samples do not cross word boundaries (vdif/payload.py:156-169)
        if bin(self.bps).count('1') != 1:
            # odd widths: single channel only, padded up to the next width
            # that divides a 32-bit word
            if tuple(self.sample_shape) != (1,):
                raise ValueError("multi-channel VDIF data requires bits per sample "
                                 "that is a power of two.")      # (the reference's words: callers match on them)
            per_word = 32 // self._bpfs
            if bin(per_word).count('1') != 1:
                raise ValueError("cannot yet sensibly handle {} data with bps={}"
                                 .format('complex' if self.complex_data else 'real', bps))
            self._bpfs = 32 // per_word

    def _decode(self, byte_start, byte_stop):
        if self.bps not in (1, 2, 4, 8) or (
                self._coder_id == _lib.CODER_MARK5B and self.bps > 2):
            raise KeyError(self.bps)
        return super()._decode(byte_start, byte_stop)

    @classmethod
    def _encode_device(cls, data, bps, edv=None, **kwargs):
        coder = _lib.CODER_MARK5B if edv == 0xab else _lib.CODER_VDIF
        return super()._encode_device(data, bps, coder_id=coder)

    @classmethod
    def fromdata(cls, data, header=None, bps=2, edv=None):
        if header is not None:
            edv = header.edv
        if edv == 0xab:                                     # Mark 5B payload in a VDIF frame
            from .. import kernels
            if data.is_complex() if hasattr(data, 'is_complex') else np.iscomplexobj(data):
                raise ValueError("Mark 5B data cannot be complex.")
            data = kernels.as_device_samples(data)
            bps = bps if header is None else header.bps
            words = cls._encode_device(data, bps, edv=edv)
            return cls(words, header, sample_shape=tuple(data.shape[1:]), bps=bps,
                       complex_data=False)
        return super().fromdata(data, header=header, bps=bps)
