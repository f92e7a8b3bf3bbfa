"""GMRT GSB format: GPU-decoded reader with the reference's call shapes."""
from .header import GSBHeader
from .payload import GSBPayload
from .frame import GSBFrame
from .base import (GSBTimeStampIO, GSBFileReader, GSBFileWriter, GSBStreamReader,
                   GSBStreamWriter, open)

__all__ = ['GSBHeader', 'GSBPayload', 'GSBFrame', 'GSBStreamReader',
           'GSBStreamWriter', 'GSBTimeStampIO', 'GSBFileReader', 'GSBFileWriter', 'open']


def info(name, **kwargs):
    """Information on a gsb file: format, rates, shapes, readability
    (the reference's ``gsb.info``; base/base.py:1440-1550)."""
    from ..io import _format_info
    return _format_info('gsb', name, dict(kwargs))


__all__ += ['info']
