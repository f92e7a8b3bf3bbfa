"""baseband_amd: MI355X-native decode path for radio-baseband files.

Drop-in for the hot path of mhvk/baseband -- ``open().read()``,
``Payload.fromfile`` and ``Payload.data`` -- with the frame-index scan and the
packed-sample decode running as hand-written HIP kernels (libbbdecode.so,
gfx950) behind a C ABI.  Decoded samples are device tensors.
"""
from . import _lib          # noqa: F401  (fails loudly if the library is missing)
from . import vdif, mark5b, mark4, guppi, dada, gsb

__version__ = '0.1.0'

FORMATS = ('vdif', 'mark5b', 'mark4', 'guppi', 'dada', 'gsb')


def open(name, mode='rs', format=None, **kwargs):
    """``baseband.open`` look-alike (io/__init__.py:178-231): dispatch to the
    opener of `format` (required: there is no format auto-detection here)."""
    if format is None:
        raise ValueError("pass format=... (one of {}); format auto-detection "
                         "is outside the decode hot path".format(FORMATS))
    try:
        module = globals()[format]
    except KeyError:
        raise ValueError("unknown format {!r}".format(format)) from None
    return module.open(name, mode, **kwargs)
