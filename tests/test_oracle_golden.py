"""The CPU oracle against the golden vectors captured from the reference
(oracle/gen_golden.py) -- runs without a GPU and without the reference."""
import hashlib

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import load_expected, load_file, bits_equal


def test_level_tables(levels_json):
    lj = levels_json
    assert np.array_equal(orc.LEVELS_1.view(np.uint32), lj['decoder_levels_1'])
    assert np.array_equal(orc.LEVELS_2.view(np.uint32), lj['decoder_levels_2'])
    assert np.array_equal(orc.LEVELS_4.view(np.uint32), lj['decoder_levels_4'])
    assert np.array_equal(orc.code_levels('vdif', 8).view(np.uint32), lj['decode_8bit'])
    for bps, key in ((1, 'vdif_lut1bit'), (2, 'vdif_lut2bit'), (4, 'vdif_lut4bit')):
        assert np.array_equal(orc.byte_lut('vdif', bps).view(np.uint32), lj[key])
    for bps, key in ((1, 'mark5b_lut1bit'), (2, 'mark5b_lut2bit')):
        assert np.array_equal(orc.byte_lut('mark5b', bps).view(np.uint32), lj[key])
    assert np.array_equal(orc.byte_lut('int', 4).view(np.uint32), lj['gsb_decode_4bit'])
    assert np.array_equal(orc.code_levels('int', 8).view(np.uint32), lj['gsb_decode_8bit'])


def test_known_answers_from_reference_tests():
    """Literal known answers quoted in the reference's test-suite
    (vdif/tests/test_vdif.py:323-336, vdif/payload.py:50-51)."""
    lut2 = orc.byte_lut('vdif', 2)
    assert np.all(lut2[0b10100101] == [-1., -1., 1., 1.])
    assert np.all(lut2[0x55] == -1.) and np.all(lut2[0xaa] == 1.)
    assert np.all(orc.byte_lut('vdif', 4)[0x88] == 0.)
    # first payload bytes of sample.vdif frame 0: 2a 0a 7c ...
    # (vdif/tests/test_vdif.py:21-25,381-382)
    d = orc.decode_flat(bytes([0x2a, 0x0a, 0x7c]), 'vdif', 2)
    assert d.astype(int).tolist() == [1, 1, 1, -3, 1, 1, -3, -3, -3, 3, 3, -1]
    # reciprocal multiplication is NOT bit-identical
    x = np.arange(256, dtype=np.float32)
    wrong = (x - np.float32(127.5)) * np.float32(1 / 35.5)
    assert np.count_nonzero(wrong != orc.code_levels('vdif', 8)) > 0


VDIF_CASES = ['sample_vdif', 'sample_mwa_vdif', 'sample_arochime_vdif',
              'sample_bps1_vdif', 'vdif_cfg2_small', 'vdif_cfg3_small',
              'vdif_bps1_c4', 'vdif_bps4_cplx_t2', 'vdif_bps8_real_c2',
              'vdif_bps8_cplx_t4', 'vdif_bps2_t8_c1', 'vdif_legacy_bps2',
              'vdif_bps4_t2_c1', 'vdif_invalid_fill0', 'vdif_invalid_fillm999',
              'vdif_edv_ab', 'vdif_triple']


@pytest.mark.parametrize('name', VDIF_CASES)
def test_vdif_oracle_matches_reference(manifest, name):
    case = manifest[name]
    raw = load_file(case['file'])
    if 'frame_rate' in case:
        fr = case['frame_rate']
    else:
        fr = int(round(case['sample_rate_hz'] / case['samples_per_frame']))
    out, info = orc.vdif_read(raw, frame_rate=fr,
                              fill_value=case.get('fill_value', 0.))
    exp = load_expected(name)
    assert bits_equal(out, exp)
    assert hashlib.sha256(out.tobytes()).hexdigest() == case['sha256']
    if 'thread_ids' in case:
        assert info['thread_ids'] == case['thread_ids']


def test_sample_vdif_literals(manifest):
    """vdif/tests/test_vdif.py:546-552,930-931: thread 0 and 3 first values."""
    exp = load_expected('sample_vdif')
    assert exp[:12, 0, 0].astype(int).tolist() == [-1, -1, 3, -1, 1, -1, 3, -1, 1, 3, -1, 1]
    assert exp[:12, 3, 0].astype(int).tolist() == [-1, 1, -1, 1, -3, -1, 3, -1, 3, -3, 1, 3]
    order = manifest['sample_vdif']['frame_order']
    assert [o[0] for o in order[:8]] == [1, 3, 5, 7, 0, 2, 4, 6]


@pytest.mark.parametrize('name', ['sample_m5b', 'm5b_c16_b2', 'm5b_c8_b1', 'm5b_c4_b2'])
def test_mark5b_oracle_matches_reference(manifest, name):
    case = manifest[name]
    raw = load_file(case['file'])
    fr = case.get('frame_rate',
                  int(round(case.get('sample_rate_hz', 0) / case['samples_per_frame'])))
    out, _ = orc.mark5b_read(raw, nchan=case['nchan'], bps=case['bps'], frame_rate=fr)
    assert bits_equal(out, load_expected(name))
    assert hashlib.sha256(out.tobytes()).hexdigest() == case['sha256']


def test_stream_masks(manifest):
    for name in ('sample_vdif', 'sample_mwa_vdif', 'sample_arochime_vdif', 'sample_bps1_vdif'):
        case = manifest[name]
        assert orc.vdif_stream_mask(case['edv']) == case['stream_mask']


M4_CASES = ['sample_m4', 'sample_32track_m4', 'sample_32track_fanout2_m4',
            'sample_16track_m4', 'sample_64track_fanout2_ft_m4',
            'm4_t64_f4', 'm4_t32_f4', 'm4_t32_f2', 'm4_t16_f4']


@pytest.mark.parametrize('name', M4_CASES)
def test_mark4_oracle_matches_reference(manifest, name):
    case = manifest[name]
    raw = load_file(case['file'])
    out, info = orc.mark4_read(raw, case['ntrack'])
    assert bits_equal(out, load_expected(name))
    assert hashlib.sha256(out.tobytes()).hexdigest() == case['sha256']
    assert info['offset0'] == case.get('offset0', 0)


def test_mark4_known_answers(manifest):
    """mark4/tests/test_mark4.py:304-308 (reorder64 of 738811025863578102) and
    :324-327,449-453 (first payload rows of sample.m4; 640 fill rows)."""
    x = np.array([738811025863578102], dtype=np.uint64)
    assert orc._m4_reorder64(x).view(np.uint8).tolist() == [118, 209, 53, 244, 148, 217, 64, 10]
    halves = x.view(np.uint32)
    assert np.array_equal(orc._m4_reorder32(halves).view(np.uint8),
                          orc._m4_reorder64(x).view(np.uint8))
    exp = load_expected('sample_m4')
    assert np.all(exp[:640] == 0.)
    assert exp[640:642].astype(int).tolist() == [[-1, 1, 1, -3, -3, -3, 1, -1],
                                                 [1, 1, -3, 1, 1, -3, -1, -1]]


def test_mark4_bitmaps_reproduce_decoders():
    """The (sign, magnitude) bit maps are equivalent to the reference's
    reorder + LUT + transpose decoders on random words."""
    import json
    from conftest import golden_path
    with open(golden_path('mark4_bitmaps.json')) as f:
        maps = json.load(f)
    assert len(maps) == 5
    for name, e in maps.items():
        nt, nchan, fanout = e['ntrack'], e['nchan'], e['fanout']
        dt = np.dtype(orc.MARK4_DTYPES[nt])
        rng = np.random.default_rng(nt + fanout)
        w = rng.integers(0, 256, size=(2000, dt.itemsize), dtype=np.uint8).view(dt).ravel()
        ref = orc.mark4_decode(w, nchan, fanout, e['signature'])
        s = np.array(e['sign_bit'], dtype=np.uint64)
        m = np.array(e['mag_bit'], dtype=np.uint64)
        w64 = w.astype(np.uint64)[:, None]
        idx = 2 * ((w64 >> s) & np.uint64(1)) + ((w64 >> m) & np.uint64(1))
        got = orc.LEVELS_2[idx.astype(int)].reshape(-1, nchan)
        assert bits_equal(np.ascontiguousarray(got), np.ascontiguousarray(ref)), name


GUPPI_CASES = ['sample_puppi', 'guppi_cf_c64_ov0', 'guppi_cf_c64_ov32',
               'guppi_tf_c8_ov16', 'guppi_cf_c6_p1', 'guppi_real_c1']
DADA_CASES = ['sample_dada', 'sample_meerkat_dada', 'sample_mkbf_dada',
              'dada_p2_c4_cplx', 'dada_p1_c1_real', 'dada_p2_c3_real']


@pytest.mark.parametrize('name', GUPPI_CASES)
def test_guppi_oracle_matches_reference(manifest, name):
    case = manifest[name]
    raw = load_file(case['file'])
    out, info = orc.guppi_read(raw)
    assert bits_equal(out, load_expected(name))
    assert info['header_nbytes'] == case['header_nbytes']
    for off, cnt, digest in case['reads']:          # partial reads incl. overlap rules
        part, _ = orc.guppi_read(raw, off, cnt)
        assert hashlib.sha256(part.tobytes()).hexdigest() == digest, (off, cnt)


def test_guppi_known_answers():
    """guppi/tests/test_guppi.py:236-249: first rows of sample_puppi.raw."""
    exp = load_expected('sample_puppi')
    assert exp.shape == (3904, 2, 4) and exp.dtype == np.complex64
    assert exp[0, 0, 0] == -7 + 12j and exp[0, 0, 1] == -32 - 10j


@pytest.mark.parametrize('name', DADA_CASES)
def test_dada_oracle_matches_reference(manifest, name):
    case = manifest[name]
    out, _ = orc.dada_read(load_file(case['file']))
    assert bits_equal(out, load_expected(name))
    for off, cnt, digest in case.get('reads', []):
        assert hashlib.sha256(np.ascontiguousarray(out[off:off + cnt]).tobytes()).hexdigest() == digest


def test_gsb_oracle_matches_reference(manifest):
    case = manifest['sample_gsb_rawdump']
    out = orc.gsb_read_rawdump(load_file(case['file']), 10, case['payload_nbytes'])
    assert bits_equal(out, load_expected('sample_gsb_rawdump'))
    # gsb/tests/test_gsb.py:235-248: 4-bit table: low nibble first, signed
    assert orc.decode_flat(bytes([0x8f, 0x70]), 'int', 4).tolist() == [-1., -8., 0., 7.]
    case = manifest['sample_gsb_phased']
    files = [[load_file(f) for f in pol] for pol in case['files']]
    out = orc.gsb_read_phased(files, 10, case['payload_nbytes'])
    exp = load_expected('sample_gsb_phased')
    assert bits_equal(out, exp)
    for off, cnt, digest in case['reads']:
        assert hashlib.sha256(np.ascontiguousarray(exp[off:off + cnt]).tobytes()).hexdigest() == digest


def test_mark4_longitudinal_crc_oracle_vs_reference():
    """the oracle's restatement of the along-track CRC-12
    check equals what the reference's crc12 (CRCStack 0x180f) finds for every
    frame of the golden cases, clean and with single header bits flipped."""
    import json
    import os
    from conftest import GOLD, load_file
    with open(os.path.join(GOLD, 'mark4_crc_cases.json')) as f:
        cases = json.load(f)['cases']
    assert len(cases) >= 10
    for c in cases:
        raw = load_file(c['file']).copy()
        for byte, bit in c['flips']:
            raw[byte] ^= np.uint8(1 << bit)
        got = orc.mark4_header_crc_bad(raw, c['offset0'], c['ntrack'], c['nframes'])
        assert [format(b, 'x') for b in got] == c['bad'], c['file']
