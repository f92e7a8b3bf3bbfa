"""Host-side sample encoders (float -> packed codes).

Only needed to synthesise input files (tests, bench, smoke); the decode
direction runs on the GPU.  Thresholds follow the reference's encoders
(base/encoding.py:63-158): 2-bit cuts at 0 and +-2.174564, 4-bit
``x*2.95 + 8.5`` clipped to 0..15, 8-bit ``rint(x*35.5 + 127.5)`` clipped to
0..255, 1-bit ``x >= 0``.
"""
import numpy as np

OPTIMAL_2BIT_HIGH = 3.316505
TWO_BIT_1_SIGMA = 2.174564
FOUR_BIT_1_SIGMA = 2.95
EIGHT_BIT_1_SIGMA = 71.0 / 2.


def codes_1bit(values):
    return (np.asarray(values) >= 0.).astype(np.uint8)


def codes_2bit(values):
    v = np.clip(np.asarray(values, dtype=np.float32), -1.5 * TWO_BIT_1_SIGMA,
                1.5 * TWO_BIT_1_SIGMA) + np.float32(2 * TWO_BIT_1_SIGMA)
    return np.floor_divide(v, np.float32(TWO_BIT_1_SIGMA)).astype(np.uint8)


def codes_4bit(values):
    v = np.asarray(values, dtype=np.float32) * np.float32(FOUR_BIT_1_SIGMA) + np.float32(8.5)
    return np.clip(v, 0., 15.).astype(np.uint8)


def codes_8bit(values):
    v = np.rint(np.asarray(values, dtype=np.float32) * np.float32(EIGHT_BIT_1_SIGMA)
                + np.float32(127.5))
    return np.clip(v, 0, 255).astype(np.uint8)


def pack_codes(codes, bps):
    """Pack an array of codes (flat, time order) LSB-first into bytes."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1)
    per = 8 // bps
    c = codes.reshape(-1, per)
    shifts = (np.arange(per, dtype=np.uint8) * bps).astype(np.uint8)
    return np.bitwise_or.reduce(c << shifts, axis=-1).astype(np.uint8)


def components(data):
    """Complex -> interleaved (re, im) float32; real passes through."""
    data = np.asarray(data)
    if data.dtype.kind == 'c':
        data = np.ascontiguousarray(data.astype(np.complex64)).view(np.float32)
    return data
