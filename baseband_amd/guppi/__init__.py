"""GUPPI/PUPPI raw format: GPU-decoded reader with the reference's call shapes."""
from .header import GUPPIHeader
from .payload import GUPPIPayload
from .frame import GUPPIFrame
from .base import GUPPIFileReader, GUPPIStreamReader, open

__all__ = ['GUPPIHeader', 'GUPPIPayload', 'GUPPIFrame', 'GUPPIFileReader',
           'GUPPIStreamReader', 'open']
