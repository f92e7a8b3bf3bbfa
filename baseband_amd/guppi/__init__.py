"""GUPPI/PUPPI raw format: GPU-decoded reader with the reference's call shapes."""
from .header import GUPPIHeader
from .payload import GUPPIPayload
from .frame import GUPPIFrame
from .base import GUPPIFileWriter, GUPPIFileReader, GUPPIStreamReader, GUPPIStreamWriter, GUPPIFileNameSequencer, open

__all__ = ['GUPPIFileWriter', 'GUPPIStreamWriter', 'GUPPIFileNameSequencer', 'GUPPIHeader', 'GUPPIPayload', 'GUPPIFrame', 'GUPPIFileReader',
           'GUPPIStreamReader', 'open']


def info(name, **kwargs):
    """Information on a guppi file: format, rates, shapes, readability
    (the reference's ``guppi.info``; base/base.py:1440-1550)."""
    from ..io import _format_info
    return _format_info('guppi', name, dict(kwargs))


__all__ += ['info']
