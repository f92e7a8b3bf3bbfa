#!/opt/conda/bin/python3.9
"""Container-side check of the ``baseband.io`` plugin seam with REAL astropy
arguments.  Run with the interpreter that has the reference's
dependencies (astropy) -- it has no torch and there is no GPU here:

    /opt/conda/bin/python3.9 tools/check_plugin_seam.py [--write-fixture]

1. `baseband_amd/base/quantities.py` is loaded BY PATH next to the real
   astropy and fed real `Quantity` / `Time` / `TimeDelta` objects; the
   conversions are compared with astropy's own (`.to_value`, `.isot`).
   ``--write-fixture`` records, for each object, the duck-typed attributes
   the module looks at and the expected result into
   tests/golden/astropy_args_cases.json -- the stand-in classes of
   tests/test_quantities.py replay exactly those on the GPU box, where astropy
   is not installed.
2. The reference's own dispatcher, ``baseband.io.open(name, 'rs',
   format='vdif_hip', sample_rate=32*u.MHz, ...)``
   (/root/reference/baseband/io/__init__.py:178), with the entry point of
   pyproject.toml registered the way an installed distribution's metadata
   does it, up to the point where the GPU is needed: the reader object comes
   back (header scan of frame 0 on the host, shape / start_time / sample_rate
   from the arguments), `seek` takes a `Time` and a `Quantity`, and `read()`
   raises the package's "needs an MI355X" error.  torch is not installed for
   this interpreter, so a STAND-IN `torch` module (no functionality: attribute
   sentinels, ``cuda.is_available() -> False``) lets the package import; none
   of the code under test touches it.
"""
import importlib.util
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path[:0] = [ROOT, REF]

# (conda's older libstdc++ would be loaded first by scipy / astropy and then does not
# satisfy libamdhip64, which the package's library links: take the system's first)
import ctypes                                       # noqa: E402
for _p in ('/usr/lib/x86_64-linux-gnu/libstdc++.so.6',):
    if os.path.exists(_p):
        ctypes.CDLL(_p, mode=ctypes.RTLD_GLOBAL)

import numpy as np                                  # noqa: E402
np.asscalar = getattr(np, 'asscalar', lambda a: a.item())   # (astropy 4.3 next to numpy 1.26 in this container)
np.alen = getattr(np, 'alen', len)
from astropy import units as u                      # noqa: E402
from astropy.time import Time, TimeDelta            # noqa: E402

spec = importlib.util.spec_from_file_location('bbq', os.path.join(ROOT, 'baseband_amd', 'base', 'quantities.py'))
q = importlib.util.module_from_spec(spec)
spec.loader.exec_module(q)


def describe(x):
    """The attributes quantities.py looks at, recorded from a real object."""
    if isinstance(x, Time):
        utc = x.utc
        return {"kind": "Time", "scale": x.scale, "repr": x.isot, "jd1": float(utc.jd1), "jd2": float(utc.jd2),
                "datetime64": str(utc.datetime64)}
    if isinstance(x, TimeDelta):
        return {"kind": "TimeDelta", "repr": repr(x), "jd1": float(x.jd1), "jd2": float(x.jd2),
                "to_value": {"s": float(x.to_value('s'))}}
    tv = {}
    for unit in ('Hz', 's', 'byte'):
        try:
            tv[unit] = float(x.to_value(unit))
        except Exception as exc:
            tv[unit] = {"raises": type(exc).__name__}
    return {"kind": "Quantity", "repr": repr(x), "to_value": tv}


LEAP_DAY_ISO = ('2016-12-31T00:00:00.000000001', '2016-12-31T12:00:00.123456789', '2016-12-31T23:59:59.999999999',
                '2017-01-01T00:00:00.000000001', '2012-06-30T18:00:00.000000375')
LEAP_ISO = ('2012-06-30T23:59:60.000000000', '2012-06-30T23:59:60.375000000', '2016-12-31T23:59:60.999999999')


def part1(write):
    cases = []

    def add(fn, x, expect):
        got = getattr(q, fn)(x)
        if isinstance(got, (np.datetime64, np.timedelta64)):
            got = str(got.astype('datetime64[ns]' if isinstance(got, np.datetime64) else 'timedelta64[ns]'))
        assert got == expect, (fn, x, got, expect)
        cases.append({"fn": fn, "arg": describe(x), "expect": expect})

    add('hz', 32 * u.MHz, 32e6)
    add('hz', 1.28e8 * u.Hz, 1.28e8)
    add('hz', 0.5 * u.GHz, 5e8)
    add('hz', 4e7 / u.s, 4e7)
    add('nbytes', 512 * u.MiB, 512 << 20)
    add('nbytes', 2 * u.Gbyte, 2 * 10 ** 9)
    add('seconds', 2.5 * u.ms, 0.0025)
    add('seconds', 3 * u.min, 180.0)
    add('seconds', TimeDelta(0.125, format='sec'), 0.125)
    add('as_timedelta', TimeDelta(86400.000000001, format='sec'), '86400000000001 nanoseconds')
    add('as_timedelta', 1250 * u.us, '1250000 nanoseconds')
    for iso, scale in (('2014-06-16T05:56:07.000000000', 'utc'), ('2014-06-16T05:56:07.123456789', 'utc'),
                       ('1999-12-31T23:59:59.999999999', 'utc'), ('2016-12-31T23:59:59.500000000', 'utc'),
                       ('2020-01-01T00:00:37.000000000', 'tai'), ('2010-03-04T05:06:07.250000000', 'tt')):
        t = Time(iso, scale=scale, precision=9)
        add('as_time', t, t.utc.isot)
    # days that end with a leap second: every instant of the day to the ns, both ways,
    # and the instant inside the leap second itself as the small type that can name it
    for iso in LEAP_DAY_ISO:
        t = Time(iso, scale='utc', precision=9)
        add('as_time', t, iso)
    for iso in LEAP_ISO:
        t = Time(iso, scale='utc', precision=9)
        leap = q.as_time(t)
        assert isinstance(leap, q.LeapSecondInstant) and str(leap) == iso, (iso, leap)
        assert q.as_time(iso) == leap
        y = t.utc.ymdhms
        cases.append({"fn": "as_time", "arg": {"kind": "Time", "scale": "utc", "repr": iso, "jd1": float(t.utc.jd1),
                                                "jd2": float(t.utc.jd2), "datetime64": None, "isot": t.utc.isot,
                                                "ymdhms": [int(y['year']), int(y['month']), int(y['day']), int(y['hour']),
                                                           int(y['minute']), float(y['second'])]},
                      "expect": iso})
    add('as_time', Time(56824.247303240743, format='mjd', precision=9),
        Time(56824.247303240743, format='mjd', precision=9).utc.isot)
    # a Time without `datetime64` (older astropy): the jd1 / jd2 route, to the ns
    for iso in ('2014-06-16T05:56:07.123456789', '2031-07-08T09:10:11.000000001'):
        t = Time(iso, scale='utc', precision=9)

        class NoD64:
            utc = types.SimpleNamespace(jd1=t.jd1, jd2=t.jd2)
            jd1, jd2 = t.jd1, t.jd2
        got = str(q.as_time(NoD64()))
        assert got == iso, (got, iso)
        cases.append({"fn": "as_time", "arg": {"kind": "Time", "scale": "utc", "repr": iso, "jd1": float(t.jd1),
                                                "jd2": float(t.jd2), "datetime64": None}, "expect": iso})
    # what must NOT convert silently
    for bad, fn in ((3 * u.m, 'hz'), (3 * u.s, 'nbytes'), (5 * u.Hz, 'seconds')):
        try:
            getattr(q, fn)(bad)
        except u.UnitConversionError:
            cases.append({"fn": fn, "arg": describe(bad), "expect": {"raises": "UnitConversionError"}})
        else:
            raise AssertionError((fn, bad))
    kw = q.normalize_kwargs(dict(sample_rate=32 * u.MHz, ref_time=Time('2014-06-16T00:00:00'), nchan=4,
                                 file_size=64 * u.MiB, subset=(1, 2), header0=None))
    assert kw == dict(sample_rate=32e6, ref_time=np.datetime64('2014-06-16T00:00:00', 'ns'), nchan=4,
                      file_size=64 << 20, subset=(1, 2), header0=None), kw
    print("part 1: {} conversions of real astropy objects agree".format(len(cases)))
    if write:
        path = os.path.join(ROOT, 'tests', 'golden', 'astropy_args_cases.json')
        import astropy
        with open(path, 'w') as f:
            json.dump({"made_by": "tools/check_plugin_seam.py --write-fixture",
                       "astropy": astropy.__version__, "numpy": np.__version__, "cases": cases}, f, indent=1)
            f.write('\n')
        print("wrote", path)


class _Anything:
    """Sentinel standing in for torch attributes (dtypes, classes)."""

    def __init__(self, name):
        self._name = name

    def __repr__(self):
        return 'torch-stand-in.' + self._name

    def __call__(self, *a, **k):
        raise RuntimeError("the torch stand-in of tools/check_plugin_seam.py was CALLED: " + self._name)

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything(self._name + '.' + name)


class _TorchStandIn(types.ModuleType):
    class Tensor:                       # isinstance(x, torch.Tensor) is False for everything here
        pass

    class device:
        def __init__(self, *a):
            self.type, self.index = 'cuda', 0

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything(name)


def part2():
    try:
        import torch                    # noqa: F401
        real_torch = True
    except ImportError:
        real_torch = False
        t = _TorchStandIn('torch')
        t.cuda = types.SimpleNamespace(is_available=lambda: False, device_count=lambda: 0,
                                       OutOfMemoryError=MemoryError)
        t.__path__ = []
        sys.modules['torch'] = t
    from importlib.metadata import EntryPoint
    import baseband_amd                 # noqa: F401  (a failure here should show itself, not "entry not loadable")
    import baseband
    import baseband.io as bio
    hasattr(bio, 'FORMATS')             # the dispatcher's first (lazy) scan of the entry points
    # what `pip install` of this repository adds to the 'baseband.io' group (pyproject.toml)
    for name in ('vdif', 'mark5b', 'mark4', 'guppi', 'dada', 'gsb'):
        bio._entries[name + '_hip'] = EntryPoint(name + '_hip', 'baseband_amd.plugin.' + name, 'baseband.io')
    # instants on days that end with a leap second, and inside the leap second, come back as the
    # same Time (a Julian date made from POSIX nanoseconds was up to a second early there)
    from baseband_amd.plugin._proxy import _as_Time
    from baseband_amd.base import quantities as bq
    for iso in LEAP_DAY_ISO + LEAP_ISO:
        t = Time(iso, scale='utc', precision=9)
        back = _as_Time(bq.as_time(t))
        assert isinstance(back, Time) and abs((back - t).to_value(u.ns)) < 0.01 and back.isot == iso, (iso, back.isot)
    gh = __import__('baseband_amd.guppi', fromlist=['GUPPIHeader']).GUPPIHeader.fromvalues(
        start_time=Time('2012-06-30T23:59:60.375', precision=9))
    import baseband.guppi
    rh = baseband.guppi.GUPPIHeader.fromvalues(start_time=Time("2012-06-30T23:59:60.375", precision=9))
    assert (gh['STT_IMJD'], gh['STT_SMJD']) == (rh['STT_IMJD'], rh['STT_SMJD']) and abs(gh['STT_OFFS'] - rh['STT_OFFS']) < 1e-9
    assert abs((_as_Time(gh.start_time) - rh.start_time).to_value(u.ns)) < 0.01
    sample = os.path.join(ROOT, 'tests', 'golden', 'samples', 'sample.vdif')
    fh = baseband.open(sample, 'rs', format='vdif_hip', sample_rate=32 * u.MHz)
    ref = baseband.open(sample, 'rs', format='vdif', sample_rate=32 * u.MHz)
    def same(ours, theirs):
        """Equal instants; through the plugin modules ours is a Time too."""
        if isinstance(ours, Time):
            return abs((ours - theirs).to_value(u.ns)) < 0.01       # (the reference adds offset / rate in day fractions)
        return str(ours) == Time(theirs, precision=9).utc.isot

    assert type(fh).__name__ == 'ReferenceTyped' and type(fh._wrapped).__module__ == 'baseband_amd.vdif.base', type(fh)
    assert fh.shape == ref.shape == (40000, 8)
    # the reference's TYPES come back at this seam (baseband_amd/plugin/_proxy.py)
    assert isinstance(fh.start_time, Time) and isinstance(fh.sample_rate, u.Quantity) and isinstance(fh.tell('time'), Time)
    assert fh.sample_rate == ref.sample_rate == 32 * u.MHz
    assert same(fh.start_time, ref.start_time) and fh.start_time.utc.isot[:19] == '2014-06-16T05:56:07'
    assert same(fh.stop_time, ref.stop_time), (fh.stop_time.isot, ref.stop_time.isot)
    assert same(fh.time, ref.time), (fh.time, ref.time)
    assert fh._wrapped.sample_rate == 32e6 and str(fh._wrapped.start_time) == '2014-06-16T05:56:07.000000000'   # plain inside
    # seek with the reference's argument types lands where the reference lands
    for target in (ref.start_time + 0.5 * u.ms, ref.start_time + TimeDelta(1e-3, format='sec')):
        assert fh.seek(target) == ref.seek(target), target
    assert fh.seek(250 * u.us) == ref.seek(250 * u.us) == 8000
    assert fh.seek(-1 * u.ms, 'end') == ref.seek(-1 * u.ms, 'end') == 8000
    assert fh.seek(TimeDelta(0.0005, format='sec'), 1) == ref.seek(TimeDelta(0.0005, format='sec'), 1) == 24000
    assert abs(fh.tell(u.ms).to_value(u.ms) - ref.tell(u.ms).to_value(u.ms)) < 1e-12
    assert same(fh.tell('time'), ref.tell('time'))
    # header0 and info through the views: the reference's types and values
    h0, r0 = fh.header0, ref.header0
    assert type(h0).__name__ == 'HeaderView' and isinstance(h0.time, Time) and same(h0.time, r0.time)
    assert same(h0.get_time(), r0.get_time()) and isinstance(h0.sample_rate, u.Quantity)
    assert h0.sample_rate == r0.sample_rate and h0.frame_rate == r0.frame_rate and h0.edv == r0.edv
    assert same(h0.ref_time, r0.ref_time) and h0['frame_nr'] == r0['frame_nr'] and h0.station == r0.station
    assert list(h0.keys()) == list(r0.keys()) and all(h0[k] == r0[k] for k in r0.keys())
    ours_info, ref_info = fh.info, ref.info
    assert isinstance(ours_info.start_time, Time) and same(ours_info.start_time, ref_info.start_time)
    assert same(ours_info.stop_time, ref_info.stop_time) and ours_info.sample_rate == ref_info.sample_rate
    assert isinstance(ours_info()['start_time'], Time) and ours_info()['sample_rate'] == 32 * u.MHz
    assert ours_info.format == ref_info.format and ours_info.shape == ref_info.shape and ours_info.bps == ref_info.bps
    fi, rfi = ours_info.file_info, ref_info.file_info
    assert fi.frame_rate == rfi.frame_rate and isinstance(fi.frame_rate, u.Quantity) and same(fi.start_time, rfi.start_time)
    assert fi.number_of_frames == rfi.number_of_frames and fi.thread_ids == rfi.thread_ids and fi.edv == rfi.edv
    # a binary reader through the dispatcher: headers typed, frame rate a Quantity
    fb = baseband.open(sample, 'rb', format='vdif_hip')
    rb = baseband.open(sample, 'rb', format='vdif')
    assert type(fb).__name__ == 'FileReaderView'
    hb, hr = fb.read_header(), rb.read_header()
    assert same(hb.time, hr.time) and fb.tell() == rb.tell() == 32
    assert fb.get_frame_rate() == rb.get_frame_rate() and isinstance(fb.get_frame_rate(), u.Quantity)
    fb.seek(5032 * 3 + 17)
    rb.seek(5032 * 3 + 17)
    assert fb.find_header(forward=True)['thread_id'] == rb.find_header(forward=True)['thread_id'] and fb.tell() == rb.tell()
    assert same(fb.info.start_time, rb.info.start_time) and fb.info.frame_rate == rb.info.frame_rate
    fb.close()
    rb.close()
    try:
        fh.read(16)
    except RuntimeError as exc:
        assert 'needs an MI355X' in str(exc), exc
        where = str(exc)
    else:
        assert real_torch, "read() returned without a GPU?"
        where = "read() ran (a GPU is present)"
    fh.close()
    ref.close()
    # Mark 5B: kday from a Time ref_time; Mark 4: decade from ref_time
    m5 = baseband.open(os.path.join(ROOT, 'tests', 'golden', 'samples', 'sample.m5b'), 'rs', format='mark5b_hip',
                       sample_rate=32 * u.MHz, nchan=8, bps=2, ref_time=Time('2014-06-13T12:00:00'))
    r5 = baseband.open(os.path.join(ROOT, 'tests', 'golden', 'samples', 'sample.m5b'), 'rs', format='mark5b',
                       sample_rate=32 * u.MHz, nchan=8, bps=2, ref_time=Time('2014-06-13T12:00:00'))
    assert same(m5.start_time, r5.start_time) and m5.shape == r5.shape
    m5.close()
    r5.close()
    m4 = baseband.open(os.path.join(ROOT, 'tests', 'golden', 'samples', 'sample.m4'), 'rs', format='mark4_hip',
                       sample_rate=32 * u.MHz, ntrack=64, ref_time=Time('2013-01-01'))
    r4 = baseband.open(os.path.join(ROOT, 'tests', 'golden', 'samples', 'sample.m4'), 'rs', format='mark4',
                       sample_rate=32 * u.MHz, ntrack=64, ref_time=Time('2013-01-01'))
    assert same(m4.start_time, r4.start_time) and m4.shape == r4.shape
    m4.close()
    r4.close()
    # the block formats and GSB through the same dispatcher
    S = os.path.join(ROOT, 'tests', 'golden', 'samples')
    for fmt, name, kw in (('dada', 'sample.dada', {}), ('guppi', 'sample_puppi.raw', {}),
                          ('gsb', os.path.join('gsb', 'sample_gsb_rawdump.timestamp'),
                           dict(raw=os.path.join(S, 'gsb', 'sample_gsb_rawdump.dat'), sample_rate=(100. / 3.) * u.MHz)),
                          ('gsb', os.path.join('gsb', 'sample_gsb_phased.timestamp'),
                           dict(raw=[[os.path.join(S, 'gsb', 'sample_gsb_phased.Pol-{}{}.dat'.format(p, k)) for k in (1, 2)]
                                     for p in ('L', 'R')], sample_rate=(100. / 3.) * u.MHz))):
        path = os.path.join(S, name)
        if not os.path.exists(path):
            print("  (no {} in tests/golden/samples: skipped)".format(name))
            continue
        ours = baseband.open(path, 'rs', format=fmt + '_hip', **kw)
        theirs = baseband.open(path, 'rs', format=fmt, **kw)
        assert ours.shape == theirs.shape, (fmt, ours.shape, theirs.shape)
        assert same(ours.start_time, theirs.start_time) and same(ours.stop_time, theirs.stop_time), fmt
        assert abs((ours.sample_rate - theirs.sample_rate).to_value(u.Hz)) < 1e-6 * theirs.sample_rate.to_value(u.Hz)
        t = theirs.start_time + 10 * u.us
        assert ours.seek(t) == theirs.seek(t), fmt
        assert same(ours.header0.time, theirs.header0.time), fmt
        if fmt in ('dada', 'guppi'):
            assert same(ours.header0.start_time, theirs.header0.start_time), fmt
            assert abs((ours.header0.offset - theirs.header0.offset).to_value(u.ns)) < 0.01, fmt
            assert abs((ours.header0.sample_rate - theirs.header0.sample_rate).to_value(u.Hz)) < 1e-6, fmt
        assert same(ours.info.start_time, theirs.info.start_time), fmt
        ours.close()
        theirs.close()
    # a writer: header keywords with a Time and a Quantity through the dispatcher
    import io as _io
    buf = _io.BytesIO()
    fw = baseband.open(buf, 'ws', format='vdif_hip', sample_rate=16 * u.MHz, nthread=2, nchan=1, bps=2,
                       complex_data=False, samples_per_frame=16000, station='me', edv=1,
                       time=Time('2018-01-02T03:04:05'))
    assert fw.sample_rate == 16 * u.MHz and fw.start_time.utc.isot[:19] == '2018-01-02T03:04:05'
    assert fw.header0.sample_rate == 16 * u.MHz and fw._wrapped.header0.sample_rate == 16e6
    # writers handed the REFERENCE's header objects as header0 (what its callers have in hand)
    from baseband import vdif as rvdif, mark5b as rm5b, mark4 as rm4, dada as rdada, guppi as rguppi
    with rvdif.open(sample, 'rs') as fr:
        rh = fr.header0
    fw = baseband.open(_io.BytesIO(), 'ws', format='vdif_hip', header0=rh, sample_rate=32 * u.MHz, nthread=8)
    assert [int(w) for w in fw.header0.words] == [int(w) for w in rh.words] and fw.header0.edv == rh.edv
    assert type(fw._wrapped.header0).__module__ == 'baseband_amd.vdif.header' and type(fw).__name__ == 'ReferenceTyped'
    with rm5b.open(os.path.join(S, 'sample.m5b'), 'rs', nchan=8, bps=2, kday=56000, sample_rate=32 * u.MHz) as fr:
        rh = fr.header0
    fw = baseband.open(_io.BytesIO(), 'ws', format='mark5b_hip', header0=rh, sample_rate=32 * u.MHz, nchan=8, bps=2)
    assert [int(w) for w in fw.header0.words] == [int(w) for w in rh.words] and same(fw.start_time, rh.time)
    with rm4.open(os.path.join(S, 'sample.m4'), 'rs', ntrack=64, decade=2010, sample_rate=32 * u.MHz) as fr:
        rh = fr.header0
    fw = baseband.open(_io.BytesIO(), 'ws', format='mark4_hip', header0=rh, sample_rate=32 * u.MHz)
    assert np.array_equal(np.asarray(fw.header0.words), np.asarray(rh.words)) and same(fw.start_time, rh.time)
    with rdada.open(os.path.join(S, 'sample.dada'), 'rs') as fr:
        rh = fr.header0
    fw = baseband.open(_io.BytesIO(), 'ws', format='dada_hip', header0=rh)
    assert fw.header0['NBIT'] == rh['NBIT'] and fw.header0.payload_nbytes == rh.payload_nbytes
    assert fw.sample_rate == rh.sample_rate and same(fw.start_time, rh.time)
    with rguppi.open(os.path.join(S, 'sample_puppi.raw'), 'rs') as fr:
        rh = fr.header0.copy()
    rh['OVERLAP'] = 0                    # (neither writer takes overlapping frames)
    fw = baseband.open(_io.BytesIO(), 'ws', format='guppi_hip', header0=rh)
    assert fw.header0['NBITS'] == rh['NBITS'] and fw.header0.payload_nbytes == rh.payload_nbytes
    assert abs((fw.sample_rate - rh.sample_rate).to_value(u.Hz)) < 1e-6 and fw.header0['SRC_NAME'] == rh['SRC_NAME']
    print("part 2: baseband.open(format='vdif_hip' / 'mark5b_hip' / 'mark4_hip' / 'dada_hip' / 'guppi_hip' / 'gsb_hip', "
          "sample_rate=32*u.MHz, ref_time=Time) returned this package's readers behind baseband_amd.plugin's views: "
          "start_time / stop_time / time / tell('time') are Time, sample_rate a Quantity, equal to the reference's "
          "(instants to 0.01 ns); header0 (time, get_time, ref_time, sample_rate, frame_rate, offset, every key), "
          "info / info() / info.file_info and an 'rb' reader's read_header / find_header / get_frame_rate answer "
          "with the reference's types and values; shapes, seek / tell agree; writers accept the reference's own header objects as "
          "header0; read() -> " + where)


if __name__ == '__main__':
    part1('--write-fixture' in sys.argv)
    part2()
    print("plugin seam ok")
