"""Mark 5B format: GPU-decoded reader with the reference's call shapes."""
from .header import Mark5BHeader
from .payload import Mark5BPayload
from .frame import Mark5BFrame
from .base import Mark5BStreamWriter, Mark5BFileWriter, Mark5BFileReader, Mark5BStreamReader, open

__all__ = ['Mark5BStreamWriter', 'Mark5BFileWriter', 'Mark5BHeader', 'Mark5BPayload', 'Mark5BFrame',
           'Mark5BFileReader', 'Mark5BStreamReader', 'open']


def info(name, **kwargs):
    """Information on a mark5b file: format, rates, shapes, readability
    (the reference's ``mark5b.info``; base/base.py:1440-1550)."""
    from ..io import _format_info
    return _format_info('mark5b', name, dict(kwargs))


__all__ += ['info']
