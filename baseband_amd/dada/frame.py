"""DADA frames (dada/frame.py): header + payload, always valid."""
from ..base.frame import FrameBase
from .header import DADAHeader
from .payload import DADAPayload

__all__ = ['DADAFrame']


class DADAFrame(FrameBase):
    _header_class = DADAHeader
    _payload_class = DADAPayload

    @classmethod
    def fromfile(cls, fh, memmap=True, verify=True):
        header = DADAHeader.fromfile(fh, verify=verify)
        payload = DADAPayload.fromfile(fh, header=header, memmap=memmap)
        return cls(header, payload, verify=verify)

    @classmethod
    def fromdata(cls, data, header, verify=True):
        payload = DADAPayload.fromdata(data, header=header)
        return cls(header, payload, verify=verify)
