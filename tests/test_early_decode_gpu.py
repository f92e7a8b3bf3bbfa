"""bb_vdif_read_window_early: the decode of a single-thread, fixed-stride VDIF window is
launched BEFORE its scan (every frame minding its own invalid-data bit,
bb_decode_params.hdr_back) and repeated through the index only when the scan's verdict
is bad.  Whatever the file holds, the result must be what the three launches in order
give (BB_EARLY_DECODE=0) and what the oracle decodes: clean files, frames flagged
invalid, a damaged sync word, a frame out of place -- 2- and 4-bit, legacy headers too.
Reference semantics: base/base.py:1083-1125 (the header check inside the read loop),
base/frame.py:191-199 (invalid frames read as fill_value)."""
import warnings

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import bits_equal

pytestmark = pytest.mark.gpu

NSETS = 2200            # x 8032 bytes = 17.7 MB: above the 16 MiB from which reads scan on the side stream


def _image(seed, bps, invalid=(), edv=0):
    from baseband_amd import synth
    return synth.random_vdif(seed, NSETS, bps=bps, edv=edv, payload_nbytes=8000, frame_rate=1000, invalid=invalid)


def _read(image, h0, monkeypatch, early, **kw):
    import torch
    from baseband_amd import vdif, kernels, _lib
    monkeypatch.setattr(kernels, '_EARLY_DECODE', early)
    dev = torch.from_numpy(image.copy()).cuda()
    with vdif.open(dev, 'rs', sample_rate=float(h0.samples_per_frame * 1000), **kw) as fh:
        out = fh.read()
        again = fh.read(0)          # (a second call must not see the first one's state)
        assert again.shape[0] == 0
        fh.seek(3 * h0.samples_per_frame + 17)
        part = fh.read(5 * h0.samples_per_frame)        # a request that enters and leaves frames halfway
    return out.cpu().numpy(), part.cpu().numpy()


@pytest.mark.parametrize('bps,edv', [(2, 0), (4, 0), (2, False)])
def test_clean_and_flagged_frames(bps, edv, monkeypatch):
    invalid = [(0, 0), (7, 0), (1234, 0), (NSETS - 1, 0)]
    image, h0 = _image(11 + bps, bps, invalid, edv=edv)
    spf = h0.samples_per_frame
    exp, _ = orc.vdif_read(image, frame_rate=1000, fill_value=-7.5)
    exp = exp.reshape(-1)
    for s, _t in invalid:
        assert np.all(exp[s * spf:(s + 1) * spf] == np.float32(-7.5))
    a, pa = _read(image, h0, monkeypatch, True, fill_value=-7.5)
    b, pb = _read(image, h0, monkeypatch, False, fill_value=-7.5)
    assert bits_equal(a.reshape(-1), exp) and bits_equal(b.reshape(-1), exp)
    assert bits_equal(pa, pb) and bits_equal(pa.reshape(-1), exp[3 * spf + 17:8 * spf + 17])


def test_the_early_form_is_the_one_that_runs(monkeypatch):
    """On a clean file the early call is taken (one decode launch at the fixed stride, no
    second one), and not taken when switched off."""
    from baseband_amd import kernels, _lib
    image, h0 = _image(5, 2)
    calls = []
    real = _lib.lib.bb_vdif_read_window_early

    def spy(*args):
        rc = real(*args)
        calls.append(rc)
        return rc
    monkeypatch.setattr(_lib.lib, 'bb_vdif_read_window_early', spy)
    monkeypatch.setattr(kernels.lib, 'bb_vdif_read_window_early', spy, raising=False)
    _read(image, h0, monkeypatch, True)
    assert _lib.BB_OK in calls
    del calls[:]
    _read(image, h0, monkeypatch, False)
    assert not calls


@pytest.mark.parametrize('damage', ['sync', 'misplaced', 'both'])
def test_bad_verdict_repeats_the_decode_through_the_index(damage, monkeypatch):
    """verify='fix' on a damaged file: the frames that fail are fill in the end, exactly
    as without the early decode (which wrote their samples first)."""
    image, h0 = _image(23, 2)
    fn = h0.frame_nbytes
    w = image.view('<u4').reshape(NSETS, fn // 4)
    if damage in ('sync', 'both'):
        w[500, 2] ^= 0x10                       # the frame length (word 2): the same in every header of a stream
    if damage in ('misplaced', 'both'):
        w[900, 1] = (w[900, 1] & 0xff000000) | ((int(w[900, 1]) & 0xffffff) + 3)      # frame_nr three too high
    outs = []
    for early in (True, False):
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            outs.append(_read(image, h0, monkeypatch, early, verify='fix'))
        assert any('problem loading frame' in str(c.message) for c in caught)
    (a, pa), (b, pb) = outs
    assert bits_equal(a, b) and bits_equal(pa, pb)
    spf = h0.samples_per_frame
    bad = [500] if damage == 'sync' else [900] if damage == 'misplaced' else [500, 900]
    for k in bad:
        assert np.all(a.reshape(-1)[k * spf:(k + 1) * spf] == 0)
    good = orc.decode_flat(image.reshape(NSETS, fn)[1, 32:], 'vdif', 2)
    assert bits_equal(a.reshape(-1)[spf:2 * spf], good)
    # verify=True refuses the same file either way
    for early in (True, False):
        with pytest.raises(ValueError):
            _read(image, h0, monkeypatch, early, verify=True)


def test_hdr_back_through_the_c_abi():
    """bb_decode_frames with hdr_back: frames whose top header bit is set decode as fill;
    launches that cannot honour it say BB_ENOTSUP instead of ignoring it."""
    import ctypes as C
    import torch
    from baseband_amd import kernels, _lib
    image, h0 = _image(3, 2, invalid=[(2, 0), (9, 0)])
    n = 16
    dev = torch.from_numpy(image[:n * 8032].copy()).cuda()
    out = torch.empty(n * 32000, dtype=torch.float32, device='cuda')
    p = _lib.DecodeParams()
    p.coder, p.bps, p.chunk, p.nslot = _lib.CODER_VDIF, 2, 1, 1
    p.payload_nbytes, p.src0, p.src_stride = 8000, 32, 8032
    p.fill_re, p.hdr_back = 42.0, 32
    rc = _lib.lib.bb_decode_frames(C.c_void_p(dev.data_ptr()), dev.numel(), None, n, C.byref(p),
                                   C.c_void_p(out.data_ptr()), out.numel(), None)
    assert rc == _lib.BB_OK
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(n, 32000)
    for k in range(n):
        want = np.full(32000, 42.0, np.float32) if k in (2, 9) else \
            orc.decode_flat(image[k * 8032 + 32:(k + 1) * 8032], 'vdif', 2)
        assert bits_equal(got[k], want), k
    p.bps = 8
    rc = _lib.lib.bb_decode_frames(C.c_void_p(dev.data_ptr()), dev.numel(), None, n, C.byref(p),
                                   C.c_void_p(out.data_ptr()), out.numel(), None)
    assert rc == _lib.BB_ENOTSUP
    p.bps, p.hdr_back = 2, 36
    rc = _lib.lib.bb_decode_frames(C.c_void_p(dev.data_ptr()), dev.numel(), None, n, C.byref(p),
                                   C.c_void_p(out.data_ptr()), out.numel(), None)
    assert rc == _lib.BB_EINVAL
