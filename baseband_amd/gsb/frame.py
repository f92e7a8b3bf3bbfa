"""GSB frames: timestamp header + payload from the raw file(s)."""
from ..base.frame import FrameBase
from .header import GSBHeader
from .payload import GSBPayload

__all__ = ['GSBFrame']


class GSBFrame(FrameBase):
    _header_class = GSBHeader
    _payload_class = GSBPayload

    @classmethod
    def fromfile(cls, fh_ts, fh_raw, payload_nbytes=1 << 22, sample_shape=(1,),
                 bps=4, complex_data=False, valid=True, verify=True):
        header = GSBHeader.fromfile(fh_ts, verify=verify)
        payload = GSBPayload.fromfile(fh_raw, payload_nbytes=payload_nbytes,
                                      sample_shape=sample_shape, bps=bps,
                                      complex_data=complex_data)
        return cls(header, payload, valid=valid, verify=verify)
