"""User-defined VDIF EDVs by subclassing (the reference's metaclass registry,
vdif/header.py:39-79; its tutorial docs/tutorials/new_edv.rst), and the sample files
the other tests do not touch (sample_vlbi.vdif, sample_drao_corrupted.vdif,
sample_vegas.raw, sample_blc.raw: vdif/tests/test_vdif.py:1319-1334,
guppi/tests/test_guppi.py:833-853) -- against what the real reference answered with the
same class definitions (oracle/gen_golden_new_edv.py -> tests/golden/new_edv_cases.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, golden_path

from baseband_amd import vdif, guppi
from baseband_amd.base.header import HeaderParser
from baseband_amd.vdif import header as vh

with open(golden_path('new_edv_cases.json')) as _f:
    REF = json.load(_f)


@pytest.fixture
def registry():
    """The class table as it was: definitions made in a test do not outlive it."""
    before = dict(vh.VDIF_HEADER_CLASSES)
    yield vh.VDIF_HEADER_CLASSES
    vh.VDIF_HEADER_CLASSES.clear()
    vh.VDIF_HEADER_CLASSES.update(before)


def test_header_parser_joins_and_defaults():
    a = HeaderParser((('x', (0, 0, 8)), ('y', (0, 8, 8, 3))))
    b = HeaderParser((('y', (1, 0, 4, 1)), ('z', (2, 0, 64, 0))))
    for joined in (a | b, a + b):
        assert list(joined.keys()) == ['x', 'y', 'z'] and joined['y'] == (1, 0, 4, 1)
        assert joined.defaults == {'x': None, 'y': 1, 'z': 0} and isinstance(joined, HeaderParser)
    assert list(a.keys()) == ['x', 'y'] and a['y'] == (0, 8, 8, 3)          # operands unchanged
    with pytest.raises(ValueError):
        HeaderParser((('bad', (0, 0)),))
    assert isinstance(vh.VDIFHeader3._header_parser, HeaderParser) and 'personality' in vh.VDIFHeader3._header_parser
    assert vh.VDIFSampleRateHeader._header_parser.defaults['sync_pattern'] == 0xACABFEED


def test_a_new_edv_class_registers_itself_and_is_used(registry):
    class VDIFHeader4(vh.VDIFHeader):
        _edv = 4
        _header_parser = HeaderParser(
            (('invalid_data', (0, 31, 1, False)), ('legacy_mode', (0, 30, 1, False)), ('seconds', (0, 0, 30)),
             ('_1_30_2', (1, 30, 2, 0x0)), ('ref_epoch', (1, 24, 6)), ('frame_nr', (1, 0, 24, 0x0)),
             ('vdif_version', (2, 29, 3, 0x1)), ('lg2_nchan', (2, 24, 5)), ('frame_length', (2, 0, 24)),
             ('complex_data', (3, 31, 1)), ('bits_per_sample', (3, 26, 5)), ('thread_id', (3, 16, 10, 0x0)),
             ('station_id', (3, 0, 16)), ('edv', (4, 24, 8)), ('validity_mask_length', (4, 16, 8, 0)),
             ('sync_pattern', (5, 0, 32, 0xACABFEED)), ('validity_mask', (6, 0, 64, 0))))

    assert registry[4] is VDIFHeader4
    want = REF['edv4']
    h = vh.VDIFHeader.fromvalues(edv=4, seconds=14363767, nchan=1, samples_per_frame=1024, station=65532,
                                 bps=2, complex_data=False, thread_id=3, validity_mask_length=60,
                                 validity_mask=(1 << 59) + 1)
    assert type(h) is VDIFHeader4 and type(h).__name__ == want['class']
    assert [int(w) for w in h.words] == want['words'] and list(h.keys()) == want['keys']
    assert {k: int(h[k]) for k in h.keys()} == want['values']
    assert h['validity_mask'] == 2 ** 59 + 1 and isinstance(h['validity_mask'], np.uint64)
    assert (h.nbytes, h.samples_per_frame, h.station) == (want['nbytes'], want['samples_per_frame'], want['station'])
    # words read back come out as the registered class, also from a file and through pickling
    again = vh.VDIFHeader(h.words)
    assert type(again) is VDIFHeader4 and again == h and again.edv == 4
    import io
    import pickle
    buf = io.BytesIO()
    h.tofile(buf)
    buf.seek(0)
    assert type(vh.VDIFHeader.fromfile(buf)) is VDIFHeader4
    assert type(pickle.loads(pickle.dumps(h))) is VDIFHeader4
    # a stream of such frames is searched by the class's own invariants (sync pattern included)
    pattern, mask = h.invariant_pattern()
    assert mask[5] == 0xffffffff and pattern[5] == 0xACABFEED
    # ... and the EDV cannot be taken twice, nor left out (the reference's messages)
    with pytest.raises(ValueError) as exc:
        class Again(vh.VDIFBaseHeader):
            _edv = 4
    assert str(exc.value) == REF['duplicate'].replace('42', '4')
    with pytest.raises(ValueError) as exc:
        class NoEDV(vh.VDIFBaseHeader):
            pass
    assert str(exc.value) == REF['no_edv']


def test_a_new_edv_with_derived_properties(registry):
    class VDIFHeader4Enhanced(vh.VDIFBaseHeader):
        _edv = 42
        _header_parser = (vh.VDIFBaseHeader._header_parser
                          | HeaderParser((('validity_mask_length', (4, 16, 8, 0)),
                                          ('sync_pattern', (5, 0, 32, 0xACABFEED)),
                                          ('validity_mask', (6, 0, 64, 0)))))
        _properties = vh.VDIFBaseHeader._properties + ('validity',)

        def verify(self):
            super().verify()
            assert 1 <= self['validity_mask_length'] <= 64

        @property
        def validity(self):
            # (the tutorial's line, with atleast_1d for NumPy 2's 0-d views)
            bitmask = np.unpackbits(np.atleast_1d(self['validity_mask']).astype('>u8').view('u1'))[::-1].astype(bool)
            return bitmask[:self['validity_mask_length']]

        @validity.setter
        def validity(self, validity):
            bitmask = np.zeros(64, dtype=bool)
            bitmask[:len(validity)] = validity
            self['validity_mask_length'] = len(validity)
            self['validity_mask'] = np.packbits(bitmask[::-1]).view('>u8')

    want = REF['edv42']
    h = vh.VDIFHeader.fromvalues(edv=42, seconds=14363767, nchan=1, samples_per_frame=1024, station=65532,
                                 bps=2, complex_data=False, thread_id=3, validity=want['validity_in'])
    assert type(h) is VDIFHeader4Enhanced and [int(w) for w in h.words] == want['words']
    assert h.validity.tolist() == want['validity_out'] and int(h['validity_mask']) == want['validity_mask']
    assert h['validity_mask_length'] == want['validity_mask_length']
    with pytest.raises(AssertionError):                     # its own verify runs
        vh.VDIFHeader.fromvalues(edv=42, seconds=1, nchan=1, samples_per_frame=1024, station=1, bps=2,
                                 complex_data=False, validity_mask_length=0)
    # the tutorial's way of taking over an EDV: pop, relabel, enter
    registry.pop(42)
    VDIFHeader4Enhanced._edv = 4
    registry[4] = VDIFHeader4Enhanced
    h4 = vh.VDIFHeader.fromvalues(edv=4, seconds=14363767, nchan=1, station=65532, bps=2, complex_data=False,
                                  thread_id=3, validity=[True] * 60)
    assert isinstance(h4, VDIFHeader4Enhanced) and h4.edv == 4 and h4['edv'] == 4 and h4.validity.sum() == 60


def _drao_class():
    class DRAOVDIFHeaderEnhanced(vh.VDIFHeader0):
        _header_parser = (vh.VDIFHeader0._header_parser
                          | HeaderParser((('link', (3, 16, 4)), ('slot', (3, 20, 6)), ('eud2', (5, 0, 32)))))

        def __init__(self, words, edv=None, verify=True, **kwargs):
            super().__init__(words, verify=False, **kwargs)
            self.mutable = True
            self['bits_per_sample'] = 3

        def verify(self):
            pass
    return DRAOVDIFHeaderEnhanced


def test_replacing_a_class_reads_the_drao_file_headers(registry):
    with vdif.open(golden_path('samples/sample_drao_corrupted.vdif'), 'rb') as fh:
        with pytest.raises(AssertionError):                 # EDV 0 with data in word 5, as the tutorial says
            fh.read_header()
    assert registry.pop(0) is vh.VDIFHeader0
    cls = _drao_class()
    assert registry[0] is cls
    with vdif.open(golden_path('samples/sample_drao_corrupted.vdif'), 'rb') as fh:
        for want in REF['drao']:
            h = fh.read_header()
            assert type(h) is cls and type(h).__name__ == want['class']
            assert (h['eud2'], h['link'], h['slot']) == (want['eud2'], want['link'], want['slot'])
            assert (h.bps, h.nchan, bool(h.complex_data)) == (want['bps'], want['nchan'], want['complex_data'])
            assert (h.frame_nbytes, h.samples_per_frame) == (want['frame_nbytes'], want['samples_per_frame'])
            fh.seek(h.payload_nbytes, 1)
            assert fh.tell() == want['tell']


@pytest.mark.gpu
def test_replacing_a_class_reads_the_drao_file_frames(registry):
    registry.pop(0)
    cls = _drao_class()
    with vdif.open(golden_path('samples/sample_drao_corrupted.vdif'), 'rb') as fh:
        for want in REF['drao']:
            frame = fh.read_frame()
            assert type(frame.header) is cls
            data = frame.data.cpu().numpy()
            assert list(data.shape) == want['shape'] and str(data.dtype) == want['dtype']
            assert data[:2].view(np.float32).reshape(-1)[:8].tolist() == want['first']
            assert hashlib.sha256(np.ascontiguousarray(data).tobytes()).hexdigest() == want['sha256']


def test_sample_vlbi_vdif_metadata():
    want = REF['vlbi']
    with vdif.open(golden_path('samples/sample_vlbi.vdif'), 'rs') as fh, \
            vdif.open(golden_path('samples/sample.vdif'), 'rs') as fc:
        assert fh.sample_rate == want['sample_rate_Hz'] and list(fh.shape) == want['shape']
        assert str(fh.start_time) == want['start_time'] and fh.start_time == fc.start_time == fh.header0.time
        assert abs((fh.stop_time - fh.start_time) / np.timedelta64(1, 'ns') - want['stop_time_ns_after_start']) < 1.
        assert [int(w) for w in fh.header0.words] == want['header0_words']


def test_oracle_forms_frame_sets_as_the_reference_does():
    """sample_vlbi.vdif: half of the threads carry other seconds.  The reference puts a
    frame into the set of the first header with its frame_nr (vdif/frame.py:201-243) and
    reads the file as sample.vdif; so does the oracle (pinned to the reference's digest)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bb_oracle_np as orc
    raw = np.fromfile(golden_path('samples/sample_vlbi.vdif'), np.uint8)
    seconds = raw.reshape(-1, 5032)[:, :4].copy().view('<u4')[:, 0] & 0x3fffffff
    assert len(set(seconds.tolist())) == 2                  # the premise
    out, info = orc.vdif_read(raw)
    assert info['frame_rate'] == 1600 and out.shape == (40000, 8, 1)
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == REF['vlbi']['sha256']


@pytest.mark.gpu
def test_sample_vlbi_vdif_samples():
    with vdif.open(golden_path('samples/sample_vlbi.vdif'), 'rs') as fh, \
            vdif.open(golden_path('samples/sample.vdif'), 'rs') as fc:
        a, b = fh.read().cpu().numpy(), fc.read().cpu().numpy()
    assert np.array_equal(a, b)
    assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == REF['vlbi']['sha256']


def test_vegas_and_breakthrough_listen_headers():
    want = REF['vegas']
    with guppi.open(golden_path('samples/sample_vegas.raw'), 'rs') as fh:
        h0 = fh.header0
        assert (h0.payload_nbytes, h0.bps, bool(h0.complex_data), h0.npol, h0.nchan) == (
            want['payload_nbytes'], want['bps'], want['complex_data'], want['npol'], want['nchan'])
        assert h0.sample_rate == want['sample_rate_Hz'] and bool(h0.sideband) == want['sideband']
        assert h0.overlap == want['overlap'] and h0.offset == want['offset_s'] and h0.nbytes == want['nbytes']
        assert h0.samples_per_frame == want['samples_per_frame'] and str(fh.start_time)[:23] == want['start_time'][:23]
    want = REF['blc']
    with guppi.open(golden_path('samples/sample_blc.raw'), 'rs') as fh:
        h0 = fh.header0
        assert (h0.nbytes, h0.bps, bool(h0.complex_data), h0.npol, h0.nchan, h0.samples_per_frame) == (
            want['nbytes'], want['bps'], want['complex_data'], want['npol'], want['nchan'], want['samples_per_frame'])
        assert h0.payload_nbytes == want['payload_nbytes'] and h0.sample_rate == want['sample_rate_Hz']
        assert str(fh.start_time)[:23] == want['start_time'][:23]


def test_header_parser_as_the_reference_tests_it():
    """base/tests/test_header_parser.py::TestHeaderParser."""
    header_parser = HeaderParser((('x0_16_4', (0, 16, 4)), ('x0_31_1', (0, 31, 1, False)), ('x1_0_32', (1, 0, 32)),
                                  ('x2_0_64', (2, 0, 64, 1 << 32))))
    extra = HeaderParser((('x4_0_32', (4, 0, 32)),))
    new = header_parser + extra
    assert len(new.keys()) == 5 and len(header_parser.keys()) == 4
    new = header_parser.copy()
    assert isinstance(new, HeaderParser)
    new.update(extra)
    assert len(new.keys()) == 5 and new['x4_0_32'] == (4, 0, 32)
    with pytest.raises(TypeError):
        header_parser + {'x4_0_32': (4, 0, 32)}
    with pytest.raises(ValueError):
        header_parser.copy().update(('x4_0_32', (4, 0, 32)))
    hp = header_parser.copy()
    words = [0x12345678, 0xffff0000, 0x0, 0xffffffff]
    hp['0_2_8'] = (0, 2, 8, 5)
    assert '0_2_8' in hp and hp.defaults['0_2_8'] == 5
    assert hp.parsers['0_2_8'](words) == (words[0] >> 2) & 0xff
    hp['0_2_8'] = (0, 1, 8, 3)
    assert hp.defaults['0_2_8'] == 3 and hp.parsers['0_2_8'](words) == (words[0] >> 1) & 0xff
    hp.update({'0_2_8': (0, 3, 8, 1)})
    assert hp.defaults['0_2_8'] == 1 and hp.parsers['0_2_8'](words) == (words[0] >> 3) & 0xff
    hp2 = header_parser + HeaderParser((('0_2_8', (0, 2, 8, 4)),))
    assert hp2.parsers['0_2_8'](words) == (words[0] >> 2) & 0xff and hp2.defaults['0_2_8'] == 4
    with pytest.raises(TypeError):
        hp + {'0_2_8': (0, 2, 8, 4)}
    with pytest.raises(Exception):
        HeaderParser((('0_2_32', (0, 2, 32, 4)),))
    with pytest.raises(Exception):
        HeaderParser((('0_2_64', (0, 2, 64, 4)),))
    # the other way round: setters
    w = [0, 0, 0, 0]
    hp2.setters['x2_0_64'](w, None)
    assert w[2:] == [0, 1] and hp2.parsers['x2_0_64'](w) == 1 << 32
    hp2.setters['x0_16_4'](w, 9)
    assert w[0] == 9 << 16 and hp2.parsers['x0_31_1'](w) is False
