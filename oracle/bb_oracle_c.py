"""ctypes access to the C oracle (oracle/bb_oracle.c) -- TEST INFRASTRUCTURE."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, 'libbboracle.so')
CODERS = {'vdif': 0, 'mark5b': 1, 'int': 2}


def available():
    return os.path.exists(_PATH)


def _lib():
    lib = C.CDLL(_PATH)
    lib.orc_decode_flat.restype = C.c_int
    lib.orc_decode_flat.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    lib.orc_levels.restype = C.c_int
    lib.orc_levels.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.orc_vdif_read.restype = C.c_long
    lib.orc_vdif_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t]
    lib.orc_mark5b_read.restype = C.c_long
    lib.orc_mark5b_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_float,
                                    C.c_void_p, C.c_size_t]
    return lib


def levels(coder, bps):
    out = np.zeros(1 << bps, np.float32)
    if _lib().orc_levels(CODERS[coder], bps, out.ctypes.data):
        raise KeyError((coder, bps))
    return out


def decode_flat(raw, coder, bps):
    raw = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8).ravel())
    out = np.empty(raw.size * 8 // bps, np.float32)
    if _lib().orc_decode_flat(raw.ctypes.data, raw.size, CODERS[coder], bps, out.ctypes.data):
        raise KeyError((coder, bps))
    return out


def vdif_read(raw, *, header_nbytes, frame_nbytes, file_threads, thread_ids,
              bps, nchan, complex_data, frame_rate, nsets, coder='vdif', fill=0.):
    raw = np.ascontiguousarray(raw)
    slot = np.full(1024, -1, np.int16)
    for s, t in enumerate(thread_ids):
        slot[t] = s
    chunk = nchan * (2 if complex_data else 1)
    E = (frame_nbytes - header_nbytes) * 8 // bps
    out = np.empty(nsets * (E // chunk) * len(thread_ids) * chunk, np.float32)
    n = _lib().orc_vdif_read(raw.ctypes.data, raw.size, header_nbytes, frame_nbytes,
                             len(file_threads), slot.ctypes.data, len(thread_ids),
                             CODERS[coder], bps, chunk, int(complex_data),
                             frame_rate, fill, out.ctypes.data, nsets)
    if n < 0:
        raise ValueError("orc_vdif_read failed: %d" % n)
    if complex_data:
        out = out.view(np.complex64)
    return out.reshape(-1, len(thread_ids), nchan)


def mark5b_read(raw, *, nchan, bps, nframes, fill=0.):
    raw = np.ascontiguousarray(raw)
    out = np.empty(nframes * 10000 * 8 // bps, np.float32)
    n = _lib().orc_mark5b_read(raw.ctypes.data, raw.size, bps, fill, out.ctypes.data, nframes)
    if n < 0:
        raise ValueError("orc_mark5b_read failed: %d" % n)
    return out.reshape(-1, nchan)
