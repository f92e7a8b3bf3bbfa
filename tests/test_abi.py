"""The C-ABI library loads and exports every symbol include/bbdecode.h
declares; host-only entry points work without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    with open(os.path.join(ROOT, 'include', 'bbdecode.h')) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bb_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_all_declared_symbols():
    from baseband_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(_lib.lib, s), s
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert bound == set(syms)
    assert _lib.lib.bb_abi_version() == 1


def test_struct_layouts_match_header():
    from baseband_amd import _lib
    assert ctypes.sizeof(_lib.FrameRec) == 16
    assert ctypes.sizeof(_lib.VDIFScanParams) == 8 + 4 + 4 + 32 + 32 + 16
    assert ctypes.sizeof(_lib.Mark5BScanParams) == 24
    assert ctypes.sizeof(_lib.DecodeParams) == 56


def test_levels_host_entry_point(levels_json):
    from baseband_amd import _lib
    lj = levels_json
    assert np.array_equal(_lib.get_levels(_lib.CODER_VDIF, 1).view(np.uint32), lj['decoder_levels_1'])
    assert np.array_equal(_lib.get_levels(_lib.CODER_VDIF, 2).view(np.uint32), lj['decoder_levels_2'])
    assert np.array_equal(_lib.get_levels(_lib.CODER_VDIF, 4).view(np.uint32), lj['decoder_levels_4'])
    assert np.array_equal(_lib.get_levels(_lib.CODER_VDIF, 8).view(np.uint32), lj['decode_8bit'])
    # Mark 5B code tables reproduce the reference's byte LUTs
    for bps, key in ((1, 'mark5b_lut1bit'), (2, 'mark5b_lut2bit')):
        lev = _lib.get_levels(_lib.CODER_MARK5B, bps)
        b = np.arange(256)[:, None]
        lut = lev[(b >> np.arange(0, 8, bps)) & ((1 << bps) - 1)]
        assert np.array_equal(lut.view(np.uint32), lj[key])
    lev = _lib.get_levels(_lib.CODER_INT, 4)
    b = np.arange(256)[:, None]
    assert np.array_equal(lev[(b >> np.arange(0, 8, 4)) & 15].view(np.uint32), lj['gsb_decode_4bit'])
    assert np.array_equal(_lib.get_levels(_lib.CODER_INT, 8).view(np.uint32), lj['gsb_decode_8bit'])


def test_error_codes():
    from baseband_amd import _lib
    out = np.empty(4, np.float32)
    p = out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    assert _lib.lib.bb_get_levels(_lib.CODER_MARK5B, 4, p, 16) == _lib.BB_ENOTSUP
    assert _lib.lib.bb_get_levels(_lib.CODER_VDIF, 3, p, 16) == _lib.BB_ENOTSUP
    assert _lib.lib.bb_get_levels(_lib.CODER_VDIF, 8, p, 4) == _lib.BB_ERANGE
    with pytest.raises(KeyError):
        _lib.check(_lib.BB_ENOTSUP, 'x')
    with pytest.raises(_lib.BBError):
        _lib.check(_lib.BB_EINVAL, 'x')
    assert _lib.lib.bb_strerror(_lib.BB_ERANGE).decode().startswith('buffer')


def test_encode_thresholds_equal_the_oracle_steps():
    """The library derives the 2-bit encoder's three step positions on the host;
    they must be the exact floats at which the NumPy restatement of
    encode_2bit_base (base/encoding.py:77-102) changes code."""
    import bb_oracle_np as orc
    from baseband_amd import _lib
    thr = _lib.encode_thresholds()
    assert thr.dtype == np.float32 and thr.shape == (3,)
    for k, t in enumerate(thr):
        below = np.nextafter(t, np.float32(-np.inf))
        codes = orc.encode_codes(np.array([below, t], np.float32), 'vdif', 2)
        assert list(codes) == [k, k + 1]
    # not the naive multiples of sigma: x + 2 sigma rounds for tiny negative x
    assert thr[1] < 0 and thr[1] == np.float32(-2.3841858e-07)
    # the step description reproduces the oracle on a dense sweep around each step
    for t in thr:
        u = np.arange(-20000, 20000, dtype=np.int64) + int(np.float32(t).view(np.int32))
        x = u.astype(np.int32).view(np.float32)
        want = orc.encode_codes(x, 'vdif', 2)
        got = (x[:, None] >= thr[None, :]).sum(1)
        assert np.array_equal(got, want)
