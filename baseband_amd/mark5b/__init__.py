"""Mark 5B format: GPU-decoded reader with the reference's call shapes."""
from .header import Mark5BHeader
from .payload import Mark5BPayload
from .frame import Mark5BFrame
from .base import Mark5BStreamWriter, Mark5BFileWriter, Mark5BFileReader, Mark5BStreamReader, open

__all__ = ['Mark5BStreamWriter', 'Mark5BFileWriter', 'Mark5BHeader', 'Mark5BPayload', 'Mark5BFrame',
           'Mark5BFileReader', 'Mark5BStreamReader', 'open']
