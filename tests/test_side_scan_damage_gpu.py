"""Reads of bytes that are resident in HBM scan their headers on a side stream once a
request passes 16 MiB (base/base.py `_scan_stream_for`).  A single-thread stream of that
size with frames flagged invalid, a header whose invariant bits are damaged and a frame
out of place must come back as the oracle decodes it / with the failed frames as fill,
whole and for requests that enter and leave frames halfway -- 2- and 4-bit, legacy
headers too.  (Written for round 6's early-decode experiment, profiles/r06_early_decode_ab.log;
the experiment went, the cases stay.)  Reference semantics: base/base.py:1083-1125,
base/frame.py:191-199."""
import warnings

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import bits_equal

pytestmark = pytest.mark.gpu

NSETS = 2200            # x 8032 bytes = 17.7 MB


def _image(seed, bps, invalid=(), edv=0):
    from baseband_amd import synth
    return synth.random_vdif(seed, NSETS, bps=bps, edv=edv, payload_nbytes=8000, frame_rate=1000, invalid=invalid)


def _read(image, h0, **kw):
    import torch
    from baseband_amd import vdif
    dev = torch.from_numpy(image.copy()).cuda()
    with vdif.open(dev, 'rs', sample_rate=float(h0.samples_per_frame * 1000), **kw) as fh:
        out = fh.read()
        assert fh.read(0).shape[0] == 0
        fh.seek(3 * h0.samples_per_frame + 17)
        part = fh.read(5 * h0.samples_per_frame)
    return out.cpu().numpy(), part.cpu().numpy()


@pytest.mark.parametrize('bps,edv', [(2, 0), (4, 0), (2, False)])
def test_clean_and_flagged_frames(bps, edv):
    invalid = [(0, 0), (7, 0), (1234, 0), (NSETS - 1, 0)]
    image, h0 = _image(11 + bps, bps, invalid, edv=edv)
    spf = h0.samples_per_frame
    exp, _ = orc.vdif_read(image, frame_rate=1000, fill_value=-7.5)
    exp = exp.reshape(-1)
    for s, _t in invalid:
        assert np.all(exp[s * spf:(s + 1) * spf] == np.float32(-7.5))
    a, pa = _read(image, h0, fill_value=-7.5)
    assert bits_equal(a.reshape(-1), exp)
    assert bits_equal(pa.reshape(-1), exp[3 * spf + 17:8 * spf + 17])


@pytest.mark.parametrize('damage', ['invariant', 'misplaced', 'both'])
def test_failed_frames_are_fill_under_fix_and_refused_under_verify(damage):
    image, h0 = _image(23, 2)
    fn = h0.frame_nbytes
    w = image.view('<u4').reshape(NSETS, fn // 4)
    if damage in ('invariant', 'both'):
        w[500, 2] ^= 0x10                       # the frame length (word 2): the same in every header of a stream
    if damage in ('misplaced', 'both'):
        w[900, 1] = (w[900, 1] & 0xff000000) | ((int(w[900, 1]) & 0xffffff) + 3)      # frame_nr three too high
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        a, pa = _read(image, h0, verify='fix')
    assert any('problem loading frame' in str(c.message) for c in caught)
    spf = h0.samples_per_frame
    bad = [500] if damage == 'invariant' else [900] if damage == 'misplaced' else [500, 900]
    flat = a.reshape(-1)
    for k in bad:
        assert np.all(flat[k * spf:(k + 1) * spf] == 0)
    for k in (1, 499, 501, 899, 901, NSETS - 1):
        good = orc.decode_flat(image.reshape(NSETS, fn)[k, 32:], 'vdif', 2)
        assert bits_equal(flat[k * spf:(k + 1) * spf], good), k
    # verify=True ends with what the reference's frame-by-frame loop meets first: bytes that are
    # no header fail its header verification (AssertionError), a sound header with another frame
    # number is a wrong frame number (ValueError) -- tests/golden/refcases/damaged_streams.json
    with pytest.raises(ValueError if damage == 'misplaced' else AssertionError):
        _read(image, h0, verify=True)
