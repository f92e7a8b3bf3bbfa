"""The C oracle against the golden vectors and the NumPy oracle (no GPU)."""
import numpy as np
import pytest

import bb_oracle_np as orc
import bb_oracle_c as orcc
from conftest import load_expected, load_file, bits_equal

pytestmark = pytest.mark.skipif(not orcc.available(),
                                reason="oracle/libbboracle.so not built")

COMBOS = [('vdif', 1), ('vdif', 2), ('vdif', 4), ('vdif', 8),
          ('mark5b', 1), ('mark5b', 2), ('int', 4), ('int', 8)]


@pytest.mark.parametrize('coder,bps', COMBOS)
def test_flat_and_levels(coder, bps):
    assert bits_equal(orcc.levels(coder, bps), orc.code_levels(coder, bps).astype(np.float32))
    raw = np.random.default_rng(bps).integers(0, 256, 4096, dtype=np.uint8)
    assert bits_equal(orcc.decode_flat(raw, coder, bps), orc.decode_flat(raw, coder, bps))
    allb = np.arange(256, dtype=np.uint8)
    assert bits_equal(orcc.decode_flat(allb, coder, bps), orc.decode_flat(allb, coder, bps))


@pytest.mark.parametrize('name', ['sample_vdif', 'vdif_cfg3_small', 'vdif_bps4_cplx_t2',
                                  'vdif_legacy_bps2', 'vdif_invalid_fillm999',
                                  'vdif_bps8_cplx_t4', 'sample_bps1_vdif'])
def test_vdif_read(manifest, name):
    case = manifest[name]
    raw = load_file(case['file'])
    h = orc.vdif_header_fields(raw[:32].view('<u4'))
    exp = load_expected(name)
    fr = case.get('frame_rate') or int(round(case['sample_rate_hz'] / case['samples_per_frame']))
    tids = case.get('thread_ids') or list(range(case['nthread']))
    out = orcc.vdif_read(raw, header_nbytes=h['header_nbytes'], frame_nbytes=h['frame_nbytes'],
                         file_threads=tids, thread_ids=tids, bps=h['bps'], nchan=h['nchan'],
                         complex_data=h['complex_data'], frame_rate=fr,
                         nsets=exp.shape[0] // h['samples_per_frame'],
                         fill=case.get('fill_value', 0.))
    assert bits_equal(out, exp)


@pytest.mark.parametrize('name', ['sample_m5b', 'm5b_c16_b2', 'm5b_c8_b1'])
def test_mark5b_read(manifest, name):
    case = manifest[name]
    raw = load_file(case['file'])
    exp = load_expected(name)
    out = orcc.mark5b_read(raw, nchan=case['nchan'], bps=case['bps'],
                           nframes=exp.shape[0] // case['samples_per_frame'])
    assert bits_equal(out, exp)
