"""baseband_amd: MI355X-native decode path for radio-baseband files.

Drop-in for the hot path of mhvk/baseband -- ``open().read()``,
``Payload.fromfile`` and ``Payload.data`` -- with the frame-index scan and the
packed-sample decode running as hand-written HIP kernels (libbbdecode.so,
gfx950) behind a C ABI.  Decoded samples are device tensors.
"""
from . import _lib          # noqa: F401  (fails loudly if the library is missing)
from . import vdif, mark5b, mark4, guppi, dada, gsb

__version__ = '0.1.0'

from .io import FORMATS, file_info, open   # noqa: E402,F401

__all__ = ['FORMATS', 'file_info', 'open', 'vdif', 'mark5b', 'mark4', 'guppi', 'dada', 'gsb']
