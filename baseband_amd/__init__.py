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


def asnumpy(data, out=None):
    """Decoded samples (device tensor) -> NumPy array on the host, through the
    pinned double-buffered copy of `staging.download` (what the reference
    returns from ``read()`` in the first place; here an explicit step, because
    the decoded output is 4-32 times larger than the file and usually wanted on
    the GPU)."""
    import numpy as np
    from .staging import download, download_new
    dtype = np.complex64 if data.is_complex() else np.float32
    if out is None:
        # (a new array: on pinned memory up to 1 GiB, so that loops of reads run at the link's rate)
        return download_new(data)
    if data.is_cuda and out.flags.c_contiguous and out.dtype == dtype:
        return download(data, out)
    out[...] = data.cpu().numpy()
    return out


__all__ += ['asnumpy']


def empty_output(shape, dtype=None, device=None):
    """Uninitialised device tensor for ``read(out=...)`` and for outputs of the
    caller's own: from the placement arena when there is one with room, else
    ``torch.empty`` (`baseband_amd.placement`)."""
    import torch
    from .placement import empty_output as _empty
    return _empty(shape, dtype=torch.float32 if dtype is None else dtype, device=device)


__all__ += ['empty_output']
