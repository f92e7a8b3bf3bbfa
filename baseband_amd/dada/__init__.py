"""DADA format: GPU-decoded reader with the reference's call shapes."""
from .header import DADAHeader
from .payload import DADAPayload, MKBFPayload
from .frame import DADAFrame
from .base import DADAFileWriter, DADAFileReader, DADAStreamReader, DADAStreamWriter, DADAFileNameSequencer, open

__all__ = ['DADAFileWriter', 'DADAStreamWriter', 'DADAFileNameSequencer', 'DADAHeader', 'DADAPayload', 'MKBFPayload', 'DADAFrame',
           'DADAFileReader', 'DADAStreamReader', 'open']


def info(name, **kwargs):
    """Information on a dada file: format, rates, shapes, readability
    (the reference's ``dada.info``; base/base.py:1440-1550)."""
    from ..io import _format_info
    return _format_info('dada', name, dict(kwargs))


__all__ += ['info']
