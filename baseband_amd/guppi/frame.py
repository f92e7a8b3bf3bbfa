"""GUPPI frames (guppi/frame.py): header + payload, always valid."""
from ..base.frame import FrameBase
from .header import GUPPIHeader
from .payload import GUPPIPayload

__all__ = ['GUPPIFrame']


class GUPPIFrame(FrameBase):
    _header_class = GUPPIHeader
    _payload_class = GUPPIPayload

    @classmethod
    def fromfile(cls, fh, memmap=True, verify=True):
        header = GUPPIHeader.fromfile(fh, verify=verify)
        payload = GUPPIPayload.fromfile(fh, header=header, memmap=memmap)
        return cls(header, payload, verify=verify)

    @classmethod
    def fromdata(cls, data, header, verify=True):
        payload = GUPPIPayload.fromdata(data, header=header)
        return cls(header, payload, verify=verify)
