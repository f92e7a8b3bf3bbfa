"""VDIF format: GPU-decoded reader with the reference's call shapes."""
from .header import VDIFHeader
from .payload import VDIFPayload
from .frame import VDIFFrame, VDIFFrameSet
from .base import VDIFStreamWriter, VDIFFileWriter, VDIFFileReader, VDIFStreamReader, open

__all__ = ['VDIFStreamWriter', 'VDIFFileWriter', 'VDIFHeader', 'VDIFPayload', 'VDIFFrame', 'VDIFFrameSet',
           'VDIFFileReader', 'VDIFStreamReader', 'open']


def info(name, **kwargs):
    """Information on a vdif file: format, rates, shapes, readability
    (the reference's ``vdif.info``; base/base.py:1440-1550)."""
    from ..io import _format_info
    return _format_info('vdif', name, dict(kwargs))


__all__ += ['info']
