"""The reference's module name for file and stream summaries
(base/file_info.py); the classes live in `baseband_amd.base.info`."""
from .info import FileReaderInfo, StreamReaderInfo

__all__ = ['FileReaderInfo', 'StreamReaderInfo']
