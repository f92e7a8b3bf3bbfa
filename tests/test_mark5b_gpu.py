"""Mark 5B through the drop-in API on the GPU, bit-exact vs the reference."""
import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import golden_path, load_expected, load_file, bits_equal

pytestmark = pytest.mark.gpu


def _open(case, **kw):
    from baseband_amd import mark5b
    fr = case.get('frame_rate')
    sr = fr * case['samples_per_frame'] if fr else case['sample_rate_hz']
    return mark5b.open(golden_path(case['file']), 'rs', sample_rate=sr,
                       kday=case['kday'], nchan=case['nchan'], bps=case['bps'], **kw)


@pytest.mark.parametrize('name', ['sample_m5b', 'm5b_c16_b2', 'm5b_c8_b1', 'm5b_c4_b2'])
def test_stream_read_matches_reference(manifest, name):
    case = manifest[name]
    with _open(case, squeeze=False) as fh:
        assert fh.shape == tuple(case['shape'])
        got = fh.read().cpu().numpy()
    assert bits_equal(got, load_expected(name))


def test_sample_known_answers(manifest):
    """mark5b/tests/test_mark5b.py:172-175: first rows of sample.m5b."""
    case = manifest['sample_m5b']
    with _open(case) as fh:
        d = fh.read(3).cpu().numpy().astype(int)
    assert d.tolist() == [[-3, -1, 1, -1, 3, -3, -3, 3],
                          [-3, 3, -1, 3, -1, -1, -1, 1],
                          [3, -1, 3, 3, 1, -1, 3, -1]]


def test_fill_pattern_frame_is_invalid(manifest):
    case = manifest['m5b_c16_b2']
    exp = load_expected('m5b_c16_b2')
    spf = case['samples_per_frame']
    assert np.all(exp[2 * spf:3 * spf] == 0.)
    with _open(case, fill_value=2.5) as fh:
        got = fh.read().cpu().numpy()
    want = exp.copy()
    want[2 * spf:3 * spf] = 2.5
    assert bits_equal(got, want)


def test_partial_reads_subset_frame(manifest):
    from baseband_amd import mark5b
    case = manifest['sample_m5b']
    exp = load_expected('sample_m5b')
    with _open(case, subset=[1, 6]) as fh:
        fh.seek(4990)
        got = fh.read(5020).cpu().numpy()
    assert bits_equal(got, np.ascontiguousarray(exp[4990:4990 + 5020][:, [1, 6]]))
    with mark5b.open(golden_path(case['file']), 'rb', kday=56000, nchan=8, bps=2) as fb:
        fb.find_header()
        frame = fb.read_frame()
        assert frame.valid and frame.shape == (5000, 8)
        assert bits_equal(frame.data.cpu().numpy(), exp[:5000])
        assert bits_equal(frame[10:20, 3].cpu().numpy(), np.ascontiguousarray(exp[10:20, 3]))


def test_payload_keyerror_for_unsupported_bps():
    from baseband_amd.mark5b import Mark5BPayload
    pl = Mark5BPayload(np.zeros(2500, '<u4'), sample_shape=(2,), bps=4)
    with pytest.raises(KeyError):
        pl.data
