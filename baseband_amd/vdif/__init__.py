"""VDIF format: GPU-decoded reader with the reference's call shapes."""
from .header import VDIFHeader
from .payload import VDIFPayload
from .frame import VDIFFrame, VDIFFrameSet
from .base import VDIFStreamWriter, VDIFFileWriter, VDIFFileReader, VDIFStreamReader, open

__all__ = ['VDIFStreamWriter', 'VDIFFileWriter', 'VDIFHeader', 'VDIFPayload', 'VDIFFrame', 'VDIFFrameSet',
           'VDIFFileReader', 'VDIFStreamReader', 'open']
