"""GSB frames: timestamp header + payload from the raw file(s)."""
from ..base.frame import FrameBase
from .header import GSBHeader
from .payload import GSBPayload

__all__ = ['GSBFrame']


class GSBFrame(FrameBase):
    _header_class = GSBHeader
    _payload_class = GSBPayload

    @classmethod
    def fromfile(cls, fh_ts, fh_raw, payload_nbytes=1 << 22, sample_shape=(1,), bps=4,
                 complex_data=False, valid=True, verify=True):
        # one timestamp line, one block of every raw file
        return cls(GSBHeader.fromfile(fh_ts, verify=verify),
                   GSBPayload.fromfile(fh_raw, payload_nbytes=payload_nbytes, bps=bps,
                                       sample_shape=sample_shape, complex_data=complex_data),
                   valid=valid, verify=verify)

    @classmethod
    def fromdata(cls, data, header=None, *, bps=4, valid=True, verify=True, **kwargs):
        """Frame from samples (packed on the GPU) and a timestamp header; without
        a header the keywords make one (gsb/frame.py:113-136)."""
        if header is None:
            header = GSBHeader.fromvalues(**kwargs)
        return cls(header, GSBPayload.fromdata(data, bps=bps), valid=valid, verify=verify)

    def tofile(self, fh_ts, fh_raw):
        """Timestamp line to `fh_ts`, payload to the raw file(s): one handle for
        rawdump, ``((L1, L2), (R1, R2))`` for phased data (gsb/frame.py:99-111)."""
        self.header.tofile(fh_ts)
        self.payload.tofile(fh_raw)

    @property
    def nbytes(self):
        return self.payload.nbytes
