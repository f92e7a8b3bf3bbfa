"""Mark 4 format: GPU-decoded reader with the reference's call shapes."""
from .header import Mark4Header
from .payload import Mark4Payload
from .frame import Mark4Frame
from .base import Mark4StreamWriter, Mark4FileWriter, Mark4FileReader, Mark4StreamReader, open

__all__ = ['Mark4StreamWriter', 'Mark4FileWriter', 'Mark4Header', 'Mark4Payload', 'Mark4Frame', 'Mark4FileReader',
           'Mark4StreamReader', 'open']


def info(name, **kwargs):
    """Information on a mark4 file: format, rates, shapes, readability
    (the reference's ``mark4.info``; base/base.py:1440-1550)."""
    from ..io import _format_info
    return _format_info('mark4', name, dict(kwargs))


__all__ += ['info']
