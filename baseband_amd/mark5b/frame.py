"""Mark 5B frames: header + payload, invalid when the payload is the
0x11223344 fill pattern everywhere (mark5b/frame.py:21-70)."""
from ..base.frame import FrameBase
from .header import Mark5BHeader
from .payload import Mark5BPayload

__all__ = ['Mark5BFrame']


class Mark5BFrame(FrameBase):
    _header_class = Mark5BHeader
    _payload_class = Mark5BPayload
    _fill_pattern = 0x11223344

    def __init__(self, header, payload, valid=None, verify=True):
        if valid is None:
            # invalid = every payload word is the fill pattern; real data differ
            # from it within a word or two, so a short look settles most frames
            words = payload.words
            head = words[:4]
            valid = bool((head != self._fill_pattern).any()
                         or (words[4:] != self._fill_pattern).any())
        super().__init__(header, payload, valid, verify)

    @classmethod
    def fromfile(cls, fh, *, kday=None, ref_time=None, sample_shape=(1,), bps=2, valid=None,
                 verify=True):
        return cls(Mark5BHeader.fromfile(fh, kday=kday, ref_time=ref_time, verify=verify),
                   Mark5BPayload.fromfile(fh, sample_shape=sample_shape, bps=bps),
                   valid, verify)

    @classmethod
    def fromdata(cls, data, header=None, bps=2, valid=True, verify=True, **kwargs):
        """Frame from (nsample, nchan) samples, packed on the GPU; without a
        header the keywords make one (mark5b/frame.py:102-124)."""
        if header is None:
            header = Mark5BHeader.fromvalues(verify=verify, **kwargs)
        return cls(header, Mark5BPayload.fromdata(data, bps=bps), valid=valid, verify=verify)

    def tofile(self, fh):
        """Header, then the payload -- or the fill pattern when the frame is
        not valid (mark5b/frame.py:126-133)."""
        import numpy as np
        self.header.tofile(fh)
        if self.valid:
            return self.payload.tofile(fh)
        return fh.write(np.full_like(self.payload.words, self._fill_pattern).tobytes())
