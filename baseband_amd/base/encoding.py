"""Level constants and the generic sample coders under the names of the
reference's ``baseband.base.encoding`` (base/encoding.py:12-160).

The tables are the ones compiled into ``libbbdecode`` (``bb_get_levels``), so
what this module reports is what the kernels use.  The functions run on the
GPU: ``encode_*_base`` return one unpacked code per sample (uint8 device
tensor), obtained by unpacking the output of the packing encoder kernel
(``bb_encode_flat``); ``decode_8bit``/``encode_8bit`` are the VDIF 8-bit
coder.  Host arrays are uploaded; there is no CPU arithmetic path here."""
import numpy as np
import torch

from .. import _lib, kernels

__all__ = ['OPTIMAL_2BIT_HIGH', 'TWO_BIT_1_SIGMA', 'FOUR_BIT_1_SIGMA',
           'EIGHT_BIT_1_SIGMA', 'decoder_levels',
           'encode_1bit_base', 'encode_2bit_base', 'encode_4bit_base',
           'decode_8bit', 'encode_8bit']

OPTIMAL_2BIT_HIGH = 3.316505
"""High level of a 2-bit digitizer whose low level is 1 (base/encoding.py:15)."""
TWO_BIT_1_SIGMA = 2.174564
"""Boundary between the low and the high level (base/encoding.py:47)."""
FOUR_BIT_1_SIGMA = 2.95
"""Scale of the 4-bit levels (base/encoding.py:49)."""
EIGHT_BIT_1_SIGMA = 71.0 / 2.
"""Scale of the 8-bit levels (base/encoding.py:51)."""


class _Levels(dict):
    """bits per sample -> float32 levels, read from the library on first use."""

    def __missing__(self, bps):
        if bps not in (1, 2, 4):
            raise KeyError(bps)
        self[bps] = _lib.get_levels(_lib.CODER_VDIF, bps)
        return self[bps]

    def keys(self):
        return (1, 2, 4)


decoder_levels = _Levels()


def _codes(values, bps):
    values = kernels.as_device_samples(values)
    shape = tuple(values.shape)
    flat = values.reshape(-1)
    if flat.numel() % 8:                        # the packer wants whole bytes
        flat = torch.nn.functional.pad(flat, (0, -flat.numel() % 8))
    packed = kernels.encode_flat(flat, _lib.CODER_VDIF, bps)
    if bps == 8:
        return packed[:values.numel()].reshape(shape)
    shifts = torch.arange(0, 8, bps, device=packed.device, dtype=torch.uint8)
    codes = (packed.unsqueeze(-1) >> shifts) & ((1 << bps) - 1)
    return codes.reshape(-1)[:values.numel()].reshape(shape)


def encode_1bit_base(values):
    """0 for negative, 1 for non-negative samples (base/encoding.py:63-74)."""
    return _codes(values, 1)


def encode_2bit_base(values):
    """Codes 0..3 with steps at -lv, 0, lv for lv = TWO_BIT_1_SIGMA
    (base/encoding.py:77-102)."""
    return _codes(values, 2)


def encode_4bit_base(values):
    """Codes 0..15: value * FOUR_BIT_1_SIGMA + 8.5, truncated and clipped
    (base/encoding.py:105-128)."""
    return _codes(values, 4)


def encode_8bit(values):
    """Codes 0..255: rint(value * EIGHT_BIT_1_SIGMA + 127.5), clipped
    (base/encoding.py:147-158)."""
    return _codes(values, 8)


def decode_8bit(words):
    """(u8 - 127.5) / EIGHT_BIT_1_SIGMA as float32 (base/encoding.py:131-144);
    `words` may be a NumPy array of any integer dtype or a device uint8 tensor."""
    if isinstance(words, torch.Tensor):
        dbuf = words.contiguous().view(torch.uint8).reshape(-1)
        if not dbuf.is_cuda:
            dbuf = dbuf.cuda()
    else:
        dbuf = kernels.to_device_bytes(np.ascontiguousarray(words))
    n = dbuf.numel()
    pad = -n % 4
    if pad:
        dbuf = torch.nn.functional.pad(dbuf, (0, pad))
    return kernels.decode_frames(dbuf, 1, n + pad, _lib.CODER_VDIF, 8)[:n]
