"""The C-ABI library loads and exports every symbol include/bbdecode.h
declares; host-only entry points work without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols(*headers):
    syms = set()
    for header in headers:
        with open(os.path.join(ROOT, 'include', header)) as f:
            text = f.read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        syms |= set(re.findall(r'\b(bb_[a-z0-9_]+)\s*\(', text))
    return sorted(syms)


PRODUCT_HEADERS = ('bbdecode.h', 'bbdecode_tune.h', 'bbdecode_arena.h')


def test_library_exports_all_declared_symbols():
    from baseband_amd import _lib
    assert sorted(h for h in os.listdir(os.path.join(ROOT, 'include')) if h.endswith('.h')) == \
        sorted(PRODUCT_HEADERS + ('bbdecode_exp.h',))
    syms = declared_symbols(*PRODUCT_HEADERS)
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(_lib.lib, s), s
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert bound == set(syms)
    import re
    with open(os.path.join(ROOT, 'include', 'bbdecode.h')) as f:
        declared = int(re.search(r'#define\s+BB_ABI_VERSION\s+(\d+)', f.read()).group(1))
    assert _lib.lib.bb_abi_version() == declared == 7
    # (the driver's build check asserts the same through __graft_entry__.build(): it must not pin a number of its own)
    with open(os.path.join(ROOT, '__graft_entry__.py')) as f:
        assert 'bb_abi_version() == declared' in f.read()


def test_product_library_carries_no_experiments():
    """Measurement variants, their knobs and the trace / pinning aids live in
    the experiment build only (make EXPERIMENTS=1, include/bbdecode_exp.h)."""
    from baseband_amd import _lib
    if _lib.EXPERIMENTS:
        pytest.skip('experiment build loaded')
    exp = set(declared_symbols('bbdecode_exp.h')) - set(declared_symbols(*PRODUCT_HEADERS))
    assert exp == {name for name, _, _ in _lib.EXPERIMENT_SIGNATURES}
    for s in exp:
        assert not hasattr(_lib.lib, s), s
    for knob in (_lib.TUNE_FLAT_VARIANT, _lib.TUNE_NT_STORES, _lib.TUNE_BYTE_LUT, _lib.TUNE_FRONT_GROUP):
        assert _lib.lib.bb_tune(knob, 1) == _lib.BB_EINVAL
    assert _lib.lib.bb_tune(_lib.TUNE_BLOCKS, 0) == _lib.BB_OK


def test_select_check_mirrors_the_launch_limits():
    """bb_decode_frames_select_check: the limits of the selecting decode,
    asked without a device (readers plan a subset against them)."""
    from baseband_amd import _lib

    def ask(bps, chunk, nslot, payload, nwithin, coder=_lib.CODER_VDIF):
        p = _lib.DecodeParams()
        p.coder, p.bps, p.chunk, p.nslot, p.payload_nbytes = coder, bps, chunk, nslot, payload
        return _lib.lib.bb_decode_frames_select_check(ctypes.byref(p), nwithin)

    assert ask(2, 32, 8, 8000, 4) == _lib.BB_OK
    assert ask(2, 32, 8, 8000, 4096) == _lib.BB_OK
    assert ask(2, 32, 8, 8000, 4097) == _lib.BB_EINVAL            # too many kept positions
    assert ask(2, 32, 8, 8000, 0) == _lib.BB_EINVAL
    assert ask(8, 16384, 1, 1 << 20, 16, _lib.CODER_INT) == _lib.BB_ENOTSUP   # sample wider than 16 tiles
    assert ask(8, 4096, 1, 1 << 20, 16, _lib.CODER_INT) == _lib.BB_OK
    assert ask(8, 4096, 1, 4096, 16, _lib.CODER_INT) == _lib.BB_OK           # 16 tiles hold one row
    assert ask(8, 4096, 1, 2048, 16, _lib.CODER_INT) == _lib.BB_EINVAL       # payload is not whole rows
    assert ask(2, 24, 2, 8000, 4) == _lib.BB_ENOTSUP              # chunk not a power of two
    assert ask(2, 4, 512, 8000, 4) == _lib.BB_ENOTSUP             # more slots than the staging buffer takes
    assert ask(3, 4, 1, 8000, 4) == _lib.BB_ENOTSUP               # no such sample width


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


def test_shipped_kernels_have_no_scratch_and_keep_four_waves_per_simd():
    """the product code object holds fewer than 150 kernel
    instantiations, none of them spills to scratch, and every streaming decode
    kernel stays at or below 128 VGPRs (at 164 the 2-bit byte-table kernel ran
    15 % slower: one wave per SIMD less).  Read from the device ISA that hipcc
    emits for gfx950 (tools/check_isa.py; no GPU needed)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_isa.py'), '--build'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    n, flagged = int(last.split()[0]), int(last.split()[2])
    assert flagged == 0 and 60 <= n < 150, last
    assert 'k_decode_flat_lds<2, true, 2, 8, 0, 0, true, 0>' in r.stdout
    # the measurement variants are not in the product code object
    for name in ('k_decode_flat_front', 'k_decode_flat_es', 'k_decode_flat_pipe', 'k_decode_flat_aln',
                 'k_decode_flat_span', 'k_decode_flat_elem', 'k_decode_flat2_bytes'):
        assert name not in r.stdout, name
