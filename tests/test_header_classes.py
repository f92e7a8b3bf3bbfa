"""The header CLASS surface of the reference on this path (VERDICT r4 missing 4):
per-EDV VDIF header classes (vdif/header.py:484-785 there), `Mark4TrackHeader`
(mark4/header.py:91-262) and the fits.Header views of `GUPPIHeader`
(guppi/header.py:17), against answers the real reference gave
(oracle/gen_golden_headers.py -> tests/golden/header_class_cases.json)."""
import json
import os
import pickle

import numpy as np
import pytest

from conftest import ROOT

from baseband_amd.vdif import header as vh
from baseband_amd.mark4.header import Mark4Header, Mark4TrackHeader
from baseband_amd.guppi.header import GUPPIHeader

SAMPLES = os.path.join(ROOT, 'tests', 'golden', 'samples')


@pytest.fixture(scope='module')
def ref():
    with open(os.path.join(ROOT, 'tests', 'golden', 'header_class_cases.json')) as f:
        return json.load(f)


def test_vdif_headers_come_back_as_the_class_of_their_edv(ref):
    for name, want in ref['vdif_files'].items():
        with open(os.path.join(SAMPLES, name), 'rb') as f:
            h = vh.VDIFHeader.fromfile(f)
        assert type(h).__name__ == want['class'], name
        assert [c.__name__ for c in type(h).__mro__ if c.__name__.startswith('VDIF')] == want['bases'], name
        assert isinstance(h, vh.VDIFHeader) and h.edv == want['edv']
        again = pickle.loads(pickle.dumps(h))
        assert type(again) is type(h) and tuple(again.words) == tuple(h.words)
        assert type(h.copy()) is type(h)


def test_fromvalues_builds_the_class_and_the_words_of_the_reference(ref):
    t = np.datetime64('2018-01-02T03:04:05')
    for want in ref['vdif_fromvalues']:
        edv = False if want['edv'] == 'legacy' else want['edv']
        kw = dict(bps=2, nchan=4, complex_data=False, station='me', time=t)
        if edv == 3:
            kw['frame_length'] = 629
        else:
            kw['samples_per_frame'] = 8000
        if edv in (1, 3):
            kw['sample_rate'] = 16e6
        h = vh.VDIFHeader.fromvalues(edv=edv, **kw)
        assert type(h).__name__ == want['class'], want
        assert [int(w) for w in h.words] == want['words'], want
        if want['frame_rate_Hz'] is not None:
            assert h.frame_rate == want['frame_rate_Hz']
        assert type(vh.VDIFHeader(h.words)).__name__ == want['class']       # the words alone say which class
    assert vh.VDIF_HEADER_CLASSES[3] is vh.VDIFHeader3 and vh.VDIF_HEADER_CLASSES[-1] is vh.VDIFLegacyHeader
    assert type(vh.VDIFHeader3(None, verify=False)).__name__ == 'VDIFHeader3' and vh.VDIFHeader3(None, verify=False).edv == 3
    assert issubclass(vh.VDIFMark5BHeader, vh.VDIFNoSampleRateHeader)


def test_mark4_track_header(ref):
    import baseband_amd as bb
    tr = ref['mark4_tracks']
    with bb.mark4.open(os.path.join(SAMPLES, tr['file']), 'rs', ntrack=64, decade=2010, sample_rate=32e6) as fh:
        h0 = fh.header0
    assert Mark4Header._track_header is Mark4TrackHeader
    for k, want in tr['tracks'].items():
        th = h0.track_header(int(k))
        assert isinstance(th, Mark4TrackHeader) and [int(w) for w in th.words] == want['words']
        direct = Mark4TrackHeader(want['words'], decade=2010)
        assert direct.track_id == want['track_id'] and direct.fraction == want['fraction']
        assert str(direct.time) == want['time'] == str(th.get_time())
        assert direct['bcd_track_id'] == want['bcd_track_id'] and direct['fan_out'] == want['fan_out']
        assert direct['converter_id'] == want['converter_id']
        assert Mark4TrackHeader(want['words'], ref_time=np.datetime64('2013-01-01')).decade == 2010
    b = Mark4TrackHeader(None, verify=False)
    b.update(bcd_headstack1=0x3344, bcd_headstack2=0x1122, headstack_id=1, fan_out=2, magnitude_bit=True,
             lsb_output=False, converter_id=5, system_id=108, crc=0, sync_pattern=0xffffffff, verify=False)
    b.track_id = 13
    b.time = np.datetime64('2015-03-02T04:05:06.25')
    assert [int(w) for w in b.words] == tr['built']['words'] and b.decade == tr['built']['decade']
    with pytest.raises(ValueError):
        b.fraction = 0.0003                     # not on the 1.25 ms grid
    with pytest.raises(AssertionError):
        Mark4TrackHeader([0, 0, 0, 0, 0])       # no sync pattern


def test_guppi_header_answers_like_a_fits_header(ref):
    g = ref['guppi']
    with open(os.path.join(SAMPLES, g['file']), 'rb') as f:
        h = GUPPIHeader.fromfile(f)
        n = f.tell()
        f.seek(0)
        raw = f.read(n)
    assert len(h.cards) == g['ncards']
    for (key, value, comment), want in zip(h.cards, g['first_cards']):
        assert [key, value, comment] == want
    text = h.tostring()
    assert len(text) == g['tostring_len'] and raw[:len(text)].decode('ascii') == text
    assert h.index('NBITS') == g['index_NBITS'] and h['nbits'] == g['lower_case_lookup'] and 'nbits' in h
    if g['comment_of_first_commented']:
        key, comment = g['comment_of_first_commented']
        assert h.comments[key] == comment
    again = GUPPIHeader.fromstring(text)
    assert again == h and again.comments == h.comments
    c = h.copy()
    c.set('NEWKEY', 3, 'a comment')
    assert c['NEWKEY'] == 3 and c.comments['NEWKEY'] == 'a comment' and c.index('NEWKEY') == len(c) - 1
    c.rename_keyword('NEWKEY', 'NEWER')
    assert 'NEWKEY' not in c and c['newer'] == 3
    c.remove('NEWER')
    assert 'NEWER' not in c and c.get('NEWER', 7) == 7
    with pytest.raises(KeyError):
        c.remove('NEWER')
    c.remove('NEWER', ignore_missing=True)
    with pytest.raises(TypeError):
        h.set('X', 1)                           # a header read from file is immutable
