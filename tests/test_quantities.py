"""Reference-typed arguments at the plugin seam.

Callers of the reference pass ``sample_rate`` as an astropy `Quantity`,
``ref_time`` / ``time`` / seek targets as `Time`, seek offsets as `Quantity` or
`TimeDelta` (/root/reference/baseband/vdif/base.py:422-454,
base/base.py:876-917); through ``baseband.open(..., format='vdif_hip')`` those
objects reach this package.  astropy is not installed next to torch, so the
stand-in classes below REPLAY what real astropy objects answered when
``tools/check_plugin_seam.py --write-fixture`` asked them (in the development
container, next to the reference: tests/golden/astropy_args_cases.json); the
same script drives the reference's own dispatcher with the real objects
(profiles/r05_check_plugin_seam.log).
"""
import io
import json
import os

import numpy as np
import pytest

from conftest import ROOT

import baseband_amd as bb
from baseband_amd.base import quantities as q

SAMPLES = os.path.join(ROOT, 'tests', 'golden', 'samples')


class UnitConversionError(ValueError):
    pass


class Recorded:
    """An object that answers ``to_value`` / ``jd1`` / ``jd2`` / ``utc`` the way
    the recorded astropy object did."""

    def __init__(self, arg):
        self._arg = arg
        if arg['kind'] in ('Time', 'TimeDelta'):
            self.jd1, self.jd2 = arg['jd1'], arg['jd2']
        if arg['kind'] == 'Time':
            d64 = arg.get('datetime64')
            self._utc = type('UTC', (), {'jd1': arg['jd1'], 'jd2': arg['jd2']})()
            if d64 is not None:
                self._utc.datetime64 = np.datetime64(d64, 'ns')
            if arg.get('ymdhms'):               # (recorded for instants inside a leap second, 23:59:60.f)
                self._utc.ymdhms = dict(zip(('year', 'month', 'day', 'hour', 'minute', 'second'), arg['ymdhms']))
                self._utc.isot = arg['isot']

    @property
    def utc(self):
        if self._arg['kind'] != 'Time':
            # astropy: ScaleValueError for a TimeDelta, AttributeError for a Quantity
            raise (ValueError if self._arg['kind'] == 'TimeDelta' else AttributeError)('utc')
        return self._utc

    def __getattr__(self, name):
        if name == 'to_value' and self.__dict__['_arg']['kind'] in ('Quantity', 'TimeDelta'):
            return self._to_value
        raise AttributeError(name)

    def _to_value(self, unit):
        v = self._arg['to_value'].get(unit, {"raises": "UnitConversionError"})
        if isinstance(v, dict):
            raise UnitConversionError(unit)
        return v

    def __float__(self):
        raise TypeError('only dimensionless scalar quantities can be converted to Python scalars')


def Q(value, unit):
    """Stand-in Quantity of frequency / time / size."""
    factor = {'Hz': 1., 'kHz': 1e3, 'MHz': 1e6, 'GHz': 1e9, 's': 1., 'ms': 1e-3, 'us': 1e-6,
              'byte': 1, 'MiB': 1 << 20}[unit]
    family = 'Hz' if unit.endswith('Hz') else 's' if unit in ('s', 'ms', 'us') else 'byte'
    return Recorded({"kind": "Quantity", "to_value": {family: value * factor}})


def T(iso):
    """Stand-in UTC Time without ``datetime64`` (the jd1 / jd2 route)."""
    ns = int(np.datetime64(iso, 'ns').astype(np.int64))
    days, rest = divmod(ns, 86400 * 10 ** 9)
    jd1 = 2440587.5 + days
    jd2 = rest / 86400e9
    if jd2 > 0.5:
        jd1, jd2 = jd1 + 1, jd2 - 1
    return Recorded({"kind": "Time", "jd1": jd1, "jd2": jd2, "datetime64": None})


def test_recorded_astropy_objects_convert_as_astropy_says():
    with open(os.path.join(ROOT, 'tests', 'golden', 'astropy_args_cases.json')) as f:
        cases = json.load(f)['cases']
    assert len(cases) >= 20
    kinds = set()
    for c in cases:
        x = Recorded(c['arg'])
        kinds.add((c['fn'], c['arg']['kind']))
        if isinstance(c['expect'], dict):
            with pytest.raises(ValueError):
                getattr(q, c['fn'])(x)
            continue
        got = getattr(q, c['fn'])(x)
        if isinstance(got, q.LeapSecondInstant):
            assert c['arg']['ymdhms'][5] >= 60
            got = str(got)
        elif isinstance(got, np.datetime64):
            got = str(got.astype('datetime64[ns]'))
        elif isinstance(got, np.timedelta64):
            got = str(got.astype('timedelta64[ns]'))
        assert got == c['expect'], (c, got)
    assert {('hz', 'Quantity'), ('nbytes', 'Quantity'), ('seconds', 'Quantity'), ('seconds', 'TimeDelta'),
            ('as_timedelta', 'TimeDelta'), ('as_time', 'Time')} <= kinds


def test_plain_arguments_pass_through():
    assert q.hz(32e6) == 32e6 and q.hz(np.float32(5)) == 5. and q.hz(None) is None
    assert q.nbytes(1 << 30) == 1 << 30
    t = np.datetime64('2014-06-16T05:56:07.25')
    assert q.as_time(t) == t and q.as_time(t).dtype == np.dtype('datetime64[ns]')
    assert q.as_time('2014-06-16T05:56:07') == np.datetime64('2014-06-16T05:56:07', 'ns')
    assert q.seconds(np.timedelta64(1500, 'us')) == pytest.approx(0.0015)
    assert q.seconds(2) == 2.0
    kw = dict(sample_rate=1e6, ref_time=np.datetime64('2020-01-01'), nchan=2, header0=object(), subset=[1])
    out = q.normalize_kwargs(kw)
    assert out == kw and out is not kw
    out = q.normalize_kwargs(dict(sample_rate=Q(32, 'MHz'), ref_time=T('2014-06-16T00:00:00'),
                                  file_size=Q(64, 'MiB'), squeeze=False))
    assert out == dict(sample_rate=32e6, ref_time=np.datetime64('2014-06-16T00:00:00', 'ns'),
                       file_size=64 << 20, squeeze=False)
    with pytest.raises(ValueError):
        q.normalize_kwargs(dict(sample_rate=Q(3, 's')))         # a rate that is not a frequency


def test_times_to_the_nanosecond_through_jd1_jd2():
    for iso in ('2014-06-16T05:56:07.123456789', '1999-12-31T23:59:59.999999999',
                '2031-07-08T09:10:11.000000001', '1970-01-01T00:00:00.000000000'):
        assert str(q.as_time(T(iso))) == iso


@pytest.mark.parametrize('rate', [32e6, Q(32, 'MHz'), Q(32000, 'kHz')])
def test_vdif_reader_takes_a_quantity_sample_rate(rate):
    """The call INTEGRATION.md section 4 promises: format='vdif_hip' ->
    baseband_amd.vdif.open(name, 'rs', sample_rate=32*u.MHz)."""
    with bb.vdif.open(os.path.join(SAMPLES, 'sample.vdif'), 'rs', sample_rate=rate) as fh:
        assert fh.sample_rate == 32e6 and isinstance(fh.sample_rate, float)
        assert fh.shape == (40000, 8)
        assert str(fh.start_time) == '2014-06-16T05:56:07.000000000'
        assert str(fh.stop_time) == '2014-06-16T05:56:07.001250000'
        # seek: Time (absolute), Quantity of time, TimeDelta; whence as in the reference
        assert fh.seek(T('2014-06-16T05:56:07.000500000')) == 16000
        assert fh.seek(Q(250, 'us')) == 8000
        assert fh.seek(Q(-1, 'ms'), 'end') == 8000
        td = Recorded({"kind": "TimeDelta", "jd1": 0.0, "jd2": 0.0005 / 86400, "to_value": {"s": 0.0005}})
        assert fh.seek(td, 1) == 24000
        assert str(fh.tell('time')) == '2014-06-16T05:56:07.000750000'


def test_the_format_blind_open_takes_them_too():
    fh = bb.open(os.path.join(SAMPLES, 'sample.vdif'), 'rs', sample_rate=Q(32, 'MHz'))
    assert fh.sample_rate == 32e6 and fh.shape == (40000, 8)
    fh.close()
    info = bb.file_info(os.path.join(SAMPLES, 'sample.m5b'), sample_rate=Q(32, 'MHz'), nchan=8, bps=2,
                        ref_time=T('2014-06-13T12:00:00'))
    assert info.format == 'mark5b' and str(info.start_time).startswith('2014-06-13T05:30:01')


def test_mark5b_and_mark4_take_a_time_ref_time():
    with bb.mark5b.open(os.path.join(SAMPLES, 'sample.m5b'), 'rs', sample_rate=Q(32, 'MHz'), nchan=8, bps=2,
                        ref_time=T('2014-06-13T12:00:00')) as fh:
        assert str(fh.start_time) == '2014-06-13T05:30:01.000000000' and fh.sample_rate == 32e6
    with bb.mark4.open(os.path.join(SAMPLES, 'sample.m4'), 'rs', sample_rate=Q(32, 'MHz'), ntrack=64,
                       ref_time=T('2013-01-01T00:00:00')) as fh:
        assert str(fh.start_time).startswith('2014-06-16T07:38:12.47') and fh.sample_rate == 32e6


def test_writer_header_keywords_take_them():
    buf = io.BytesIO()
    fw = bb.vdif.open(buf, 'ws', sample_rate=Q(16, 'MHz'), nthread=2, nchan=1, bps=2, complex_data=False,
                      samples_per_frame=16000, station='me', edv=1, time=T('2018-01-02T03:04:05'))
    assert fw.sample_rate == 16e6 and str(fw.start_time) == '2018-01-02T03:04:05.000000000'
    assert fw.header0.sample_rate == 16e6
    h = bb.vdif.header.VDIFHeader.fromvalues(edv=1, time=T('2018-01-02T03:04:05.5'), sample_rate=Q(16, 'MHz'),
                                             nchan=1, bps=2, complex_data=False, samples_per_frame=16000,
                                             station='me', frame_rate=Q(1, 'kHz'))
    assert str(h.get_time(frame_rate=Q(1, 'kHz'))) == '2018-01-02T03:04:05.500000000'
    d = bb.dada.header.DADAHeader.fromvalues(
        time=T('2013-07-02T01:39:20'), offset=Q(500, 'ms'), sample_rate=Q(16, 'MHz'), bps=8, complex_data=True,
        npol=2, nchan=1, payload_nbytes=64000, start_time=T('2013-07-02T01:39:19.5')) \
        if hasattr(bb.dada.header.DADAHeader, 'fromvalues') else None
    if d is not None:
        assert d.sample_rate == 16e6 and abs(d.offset - 0.5) < 1e-9


def test_gsb_takes_a_quantity_sample_rate():
    """Found by tools/check_plugin_seam.py with a real Quantity: the GSB reader
    did arithmetic on the rate before converting it."""
    d = os.path.join(SAMPLES, 'gsb')
    rate = 100e6 / 3.
    for r in (rate, Q(100. / 3., 'MHz')):
        with bb.gsb.open(os.path.join(d, 'sample_gsb_rawdump.timestamp'), 'rs',
                         raw=os.path.join(d, 'sample_gsb_rawdump.dat'), sample_rate=r) as fh:
            assert abs(fh.sample_rate - rate) < 1e-3 and fh.shape[0] > 0
            shape = fh.shape
    with bb.gsb.open(os.path.join(d, 'sample_gsb_phased.timestamp'), 'rs',
                     raw=[[os.path.join(d, 'sample_gsb_phased.Pol-{}{}.dat'.format(p, k)) for k in (1, 2)] for p in ('L', 'R')],
                     sample_rate=Q(100. / 3., 'MHz')) as fh:
        assert abs(fh.sample_rate - rate) < 1e-3 and fh.shape[1:] == (2, 512)
    assert shape[0] > 0


def test_writers_adopt_a_foreign_header0():
    """``header0=`` may be the REFERENCE's header object (what its callers pass
    through the plugin seam): anything with `words` (VDIF, Mark 5B, Mark 4), a
    mapping with `comments` (DADA), `cards` (GUPPI: a fits.Header) is rebuilt as
    this package's header.  tools/check_plugin_seam.py does it with the real
    objects; here with look-alikes made from this package's own."""
    import collections
    with open(os.path.join(SAMPLES, 'sample.vdif'), 'rb') as f:
        ours = bb.vdif.header.VDIFHeader.fromfile(f)

    class ForeignWords:
        def __init__(self, words, **attrs):
            self.words = words
            self.__dict__.update(attrs)

    fw = bb.vdif.open(io.BytesIO(), 'ws', header0=ForeignWords(tuple(ours.words), edv=ours.edv), sample_rate=Q(32, 'MHz'),
                      nthread=8)
    assert isinstance(fw.header0, bb.vdif.header.VDIFHeader3) and tuple(fw.header0.words) == tuple(ours.words)
    with bb.mark5b.open(os.path.join(SAMPLES, 'sample.m5b'), 'rs', nchan=8, bps=2, kday=56000, sample_rate=32e6) as fr:
        m5 = fr.header0
    fw = bb.mark5b.open(io.BytesIO(), 'ws', header0=ForeignWords(tuple(m5.words), kday=56000), sample_rate=32e6, nchan=8, bps=2)
    assert tuple(fw.header0.words) == tuple(m5.words) and fw.start_time == m5.time
    with bb.mark4.open(os.path.join(SAMPLES, 'sample.m4'), 'rs', ntrack=64, decade=2010, sample_rate=32e6) as fr:
        m4 = fr.header0
    fw = bb.mark4.open(io.BytesIO(), 'ws', header0=ForeignWords(np.array(m4.words), decade=2010), sample_rate=32e6)
    assert np.array_equal(fw.header0.words, m4.words) and fw.start_time == m4.time
    with bb.dada.open(os.path.join(SAMPLES, 'sample.dada'), 'rs') as fr:
        dd = fr.header0

    class ForeignDADA(collections.OrderedDict):
        comments = {'NBIT': 'bits', '_3': 'a comment line'}

    foreign = ForeignDADA(list(dd.items()) + [('_3', None)])
    fw = bb.dada.open(io.BytesIO(), 'ws', header0=foreign)
    assert isinstance(fw.header0, bb.dada.header.DADAHeader) and dict(fw.header0) == dict(dd)
    assert fw.header0.comments == {'NBIT': 'bits'}
    with open(os.path.join(SAMPLES, 'sample_puppi.raw'), 'rb') as f:
        gg = bb.guppi.header.GUPPIHeader.fromfile(f)
    Card = collections.namedtuple('Card', 'keyword value comment')

    class ForeignFits:
        def __init__(self, h):
            self.cards = [Card(k, (0 if k == 'OVERLAP' else v), c) for k, v, c in h.cards] + [Card('COMMENT', 'x', '')]
            self.frame_nbytes = h.frame_nbytes

    fw = bb.guppi.open(io.BytesIO(), 'ws', header0=ForeignFits(gg))
    assert isinstance(fw.header0, bb.guppi.header.GUPPIHeader) and fw.header0['SRC_NAME'] == gg['SRC_NAME']
    assert fw.header0['OVERLAP'] == 0 and 'COMMENT' not in fw.header0
