"""Host-side sample encoders (float -> packed codes) -- INPUT SYNTHESIS ONLY.

Used by `synth.py` to make test / smoke / bench input files on machines
without a GPU.  Nothing on the product path imports this module: payloads,
frames and stream writers encode with the GPU kernels (`bb_encode_flat`,
`bb_encode_mark4`) and fail without them.  Thresholds follow the reference's encoders
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


def encode_mark5b(comp, bps):
    """float32 components -> packed bytes.  2-bit codes are re-ordered so
    that the sign sits on the even and the magnitude on the odd bit stream
    (mark5b/payload.py:97-106); 1 bit stores the sign bit."""
    if bps == 1:
        return pack_codes(np.signbit(np.asarray(comp)).astype(np.uint8), 1)
    if bps == 2:
        reorder = np.array([0, 2, 1, 3], dtype=np.uint8)
        return pack_codes(reorder[codes_2bit(comp)], 2)
    raise ValueError(f"Mark5BPayload cannot encode data with {bps} bits")


def encode_mark4(data, header):
    """float data (nsample, nchan) -> stream words, inverse of the bit maps:
    2-bit code = 2*sign + magnitude with levels {-Hi,-1,+1,+Hi}."""
    from .mark4._bitmaps import BITMAPS
    from .mark4.header import MARK4_DTYPES
    key = (header.nchan, header.magnitude_signature() or header.bps, header.fanout)
    maps = BITMAPS[key]
    ntrack = maps['ntrack']
    dtype = np.dtype(MARK4_DTYPES[ntrack])
    codes = codes_2bit(np.asarray(data, dtype=np.float32))
    opw = ntrack // 2
    codes = codes.reshape(-1, opw).astype(np.uint64)
    sign = (codes >> np.uint64(1)) & np.uint64(1)
    mag = codes & np.uint64(1)
    words = np.zeros(codes.shape[0], dtype=np.uint64)
    for j in range(opw):
        words |= sign[:, j] << np.uint64(maps['sign_bit'][j])
        words |= mag[:, j] << np.uint64(maps['mag_bit'][j])
    return words.astype(dtype)
