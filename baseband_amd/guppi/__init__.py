"""GUPPI/PUPPI raw format: GPU-decoded reader with the reference's call shapes."""
from .header import GUPPIHeader
from .payload import GUPPIPayload
from .frame import GUPPIFrame
from .base import GUPPIFileWriter, GUPPIFileReader, GUPPIStreamReader, GUPPIStreamWriter, GUPPIFileNameSequencer, open

__all__ = ['GUPPIFileWriter', 'GUPPIStreamWriter', 'GUPPIFileNameSequencer', 'GUPPIHeader', 'GUPPIPayload', 'GUPPIFrame', 'GUPPIFileReader',
           'GUPPIStreamReader', 'open']
