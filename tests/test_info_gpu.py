"""``fh.info`` against the reference's info on the sample files
(tests/golden/info_cases.json, oracle/gen_golden.py `info`) and the
readability checks its corrupt-file tests make."""
import json

import numpy as np
import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu

with open(golden_path('info_cases.json')) as _f:
    INFO = json.load(_f)

MODULE = {'vdif': 'vdif', 'm5b': 'mark5b', 'm4': 'mark4', 'dada': 'dada', 'puppi': 'guppi'}
STREAM_RATE = {'sample_m5b': 32e6, 'sample_m4': 32e6, 'sample_m4_32': 32e6,
               'sample_mwa_vdif': 1.28e6, 'sample_arochime_vdif': 800e6 / 1024. / 2.,
               'sample_bps1_vdif': 8e6}


def _module(key):
    import importlib
    name = next(v for k, v in MODULE.items() if k in key)
    return importlib.import_module('baseband_amd.' + name)


def _same(mine, ref, key):
    if key == 'start_time' or key == 'stop_time':
        return np.datetime64(mine, 'ns') == np.datetime64(ref, 'ns')
    if isinstance(ref, float):
        return abs(mine - ref) <= 1e-9 * abs(ref)
    if isinstance(ref, list):
        return list(mine) == ref
    return mine == ref


@pytest.mark.parametrize('key', sorted(INFO))
def test_file_info_matches_reference(key):
    case = INFO[key]
    mod = _module(key)
    ref = case['file_info']
    with mod.open(golden_path('samples/' + case['file']), 'rb', **case['kwargs']) as fh:
        info = fh.info
        mine = info()
    for name, value in ref.items():
        if name in ('errors', 'warnings'):
            assert set(mine.get(name, {})) == set(value), name
        elif name == 'missing':
            assert mine['missing'] == value
        else:
            assert name in mine, name
            assert _same(mine[name], value, name), (name, mine[name], value)
    assert set(mine) - {'errors', 'warnings'} == set(ref) - {'errors', 'warnings'}
    assert bool(info) and 'information' in repr(info)


@pytest.mark.parametrize('key', sorted(k for k in INFO if 'stream_info' in INFO[k]))
def test_stream_info_matches_reference(key):
    case = INFO[key]
    mod = _module(key)
    ref = case['stream_info']
    kwargs = dict(case['kwargs'])
    if key in STREAM_RATE:
        kwargs['sample_rate'] = STREAM_RATE[key]
    with mod.open(golden_path('samples/' + case['file']), 'rs', **kwargs) as fh:
        mine = fh.info()
        assert fh.info.readable and fh.info.checks['continuous'] == 'no obvious gaps'
        assert fh.tell() == 0
    for name, value in ref.items():
        if name in ('errors', 'warnings'):
            continue
        assert name in mine, name
        if name == 'verify':
            assert str(mine[name]) == str(value)
        else:
            assert _same(mine[name], value, name), (name, mine[name], value)
    assert not fh.info.readable                      # closed stream


def test_info_of_corrupt_streams(tmp_path):
    """The readability checks of the reference's corrupt-file tests
    (mark5b/tests/test_corrupt_files.py:100-140,
    mark4/tests/test_corrupt_files.py:56-80)."""
    from baseband_amd import mark5b, mark4
    with open(golden_path('fixed_corrupt_cases.json')) as f:
        fixed = json.load(f)
    files = np.load(golden_path('fixed_corrupt_files.npz'))
    sample = open(golden_path('samples/sample.m5b'), 'rb').read()
    tail = files['m5b_sample_tail'].tobytes()
    # first byte of header 2 missing -> frames 1, 2 bad: trouble starts at frame 1
    for (lo, hi), bad_start in (((20032, 20033), 1), ((20096, 20100), 2), ((30060, 30070), 3)):
        p = tmp_path / 'c.m5b'
        p.write_bytes(sample[:lo] + sample[hi:] + tail)
        with mark5b.open(str(p), 'rs', nchan=8, bps=2, kday=56000, verify=True) as fv:
            info = fv.info
            assert not info.readable and not info.checks['continuous']
            msg = 'While reading at {}'.format(bad_start * fv.samples_per_frame)
            assert msg in str(info.errors['continuous'])
            fv.verify = 'fix'
            info = fv.info
            assert info.readable and 'fixable' in info.checks['continuous']
            assert msg in info.warnings['continuous']
            assert 'problem loading' in info.warnings['continuous']
            assert fv.tell() == 0
    fake = files['m4_fake'].tobytes()
    t0 = np.datetime64('2010-11-12T13:14:15')
    for (lo, hi), bad_start in (((40000, 80000), 1), ((120000, 200000), 3), ((78000, 82000), 1)):
        p = tmp_path / 'c.m4'
        p.write_bytes(fake[:lo] + fake[hi:])
        with mark4.open(str(p), 'rs', verify=True, sample_rate=100e3, ref_time=t0) as fv:
            assert not fv.info.readable and not fv.info.checks['continuous']
            msg = 'While reading at {}'.format(bad_start * fv.samples_per_frame)
            assert msg in str(fv.info.errors['continuous'])
        with mark4.open(str(p), 'rs', verify='fix', sample_rate=100e3, ref_time=t0) as ff:
            assert ff.info.readable and 'fixable' in ff.info.checks['continuous']
            assert msg in ff.info.warnings['continuous']
            assert 'problem loading frame' in ff.info.warnings['continuous']


# ---- format auto-detection (io/__init__.py:99-231) -------------------------
DETECT = [('samples/sample.vdif', 'vdif', {}), ('samples/sample_mwa.vdif', 'vdif', dict(sample_rate=1.28e6)),
          ('samples/sample_arochime.vdif', 'vdif', dict(sample_rate=390625.)),
          ('samples/sample_bps1.vdif', 'vdif', dict(sample_rate=8e6)),
          ('samples/sample.m5b', 'mark5b', dict(nchan=8, kday=56000)),
          ('samples/sample.m4', 'mark4', dict(decade=2010)),
          ('samples/sample_32track.m4', 'mark4', dict(ref_time=np.datetime64('2015-01-01'))),
          ('samples/sample_16track.m4', 'mark4', dict(decade=2010)),
          ('samples/sample.dada', 'dada', {}), ('samples/sample_puppi.raw', 'guppi', {})]


@pytest.mark.parametrize('path,fmt,kwargs', DETECT)
def test_format_is_detected_and_file_opens(path, fmt, kwargs):
    import importlib
    import baseband_amd
    info = baseband_amd.file_info(golden_path(path), **kwargs)
    assert info and info.format == fmt and info.readable
    assert info.used_kwargs == {k: v for k, v in kwargs.items()}
    with baseband_amd.open(golden_path(path), 'rs', **kwargs) as fh, \
            importlib.import_module('baseband_amd.' + fmt).open(golden_path(path), 'rs', **kwargs) as f1:
        assert type(fh) is type(f1) and fh.shape == f1.shape
        assert bool((fh.read() == f1.read()).all())
    with baseband_amd.open(golden_path(path), 'rb', **kwargs) as fb:
        assert type(fb).__name__.lower().startswith(fmt)


def test_detection_reports_missing_and_inconsistent_arguments(tmp_path):
    import baseband_amd
    info = baseband_amd.file_info(golden_path('samples/sample.m5b'))
    assert info.format == 'mark5b' and set(info.missing) == {'nchan', 'kday', 'ref_time'}
    with pytest.raises(TypeError, match='missing required arguments'):
        baseband_amd.open(golden_path('samples/sample.m5b'), 'rs')
    # arguments the format does not take are classified against the file
    info = baseband_amd.file_info(golden_path('samples/sample.vdif'), nchan=8,
                                  ref_time=np.datetime64('2014-01-01'), kday=56000, decade=2010)
    assert info.format == 'vdif' and not info.inconsistent_kwargs
    assert set(info.consistent_kwargs) == {'nchan', 'ref_time', 'kday', 'decade'}
    info = baseband_amd.file_info(golden_path('samples/sample.vdif'), nchan=4, decade=2000)
    assert set(info.inconsistent_kwargs) == {'nchan', 'decade'}
    with pytest.raises(ValueError, match='inconsistent'):
        baseband_amd.open(golden_path('samples/sample.vdif'), 'rs', nchan=4)
    junk = tmp_path / 'junk.bin'
    junk.write_bytes(np.random.default_rng(1).integers(0, 256, 100000, dtype=np.uint8).tobytes())
    assert not baseband_amd.file_info(str(junk))
    with pytest.raises(ValueError, match='could not be auto-determined'):
        baseband_amd.open(str(junk), 'rs')
    with pytest.raises(ValueError):
        baseband_amd.open(str(junk), 'ws', sample_rate=1e6)


# ---- stream API fuzz against the reference ----------------------------------
with open(golden_path('stream_fuzz_cases.json')) as _f:
    FUZZ = json.load(_f)
FUZZ_KW = {'samples/sample_arochime.vdif': dict(sample_rate=800e6 / 1024. / 2.),
           'samples/sample_mwa.vdif': dict(sample_rate=1.28e6),
           'samples/sample.m5b': dict(kday=56000, nchan=8, sample_rate=32e6),
           'samples/sample.m4': dict(ntrack=64, decade=2010, sample_rate=32e6),
           'samples/sample_32track_fanout2.m4': dict(ntrack=32, decade=2010)}


@pytest.mark.parametrize('i', range(len(FUZZ)),
                         ids=['%s-%d' % (c['fmt'], k) for k, c in enumerate(FUZZ)])
def test_stream_calls_match_reference(i):
    """Random squeeze / subset openings and seek + read sequences: shapes and
    digests as the reference returns them (oracle/gen_golden.py `stream_fuzz`)."""
    import hashlib
    import importlib
    case = FUZZ[i]
    mod = importlib.import_module('baseband_amd.' + case['fmt'])
    subset = tuple(slice(v[1], v[2], v[3]) if isinstance(v, list) and v and v[0] == 'slice' else v
                   for v in case['subset'])
    kwargs = dict(FUZZ_KW.get(case['file'], {}), squeeze=case['squeeze'], subset=subset)
    if case['fmt'] == 'gsb':
        d = golden_path('samples/gsb/')
        if 'rawdump' in case['file']:
            kwargs.update(raw=d + 'sample_gsb_rawdump.dat', samples_per_frame=8192)
        else:
            kwargs.update(raw=[[d + 'sample_gsb_phased.Pol-%s%d.dat' % (p, k) for k in (1, 2)] for p in 'LR'],
                          samples_per_frame=8)
    if 'error' in case:
        with pytest.raises(Exception):
            with mod.open(golden_path(case['file']), 'rs', **kwargs) as fh:
                fh.read(1)
        return
    with mod.open(golden_path(case['file']), 'rs', **kwargs) as fh:
        assert list(fh.shape) == case['shape'] and list(fh.sample_shape) == case['sample_shape']
        for op in case['ops']:
            fh.seek(op['seek'])
            d = fh.read(op['count']).cpu().numpy()
            assert list(d.shape) == op['shape'] and fh.tell() == op['tell']
            assert hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest() == op['sha256']


# ---- item access fuzz (payload / frame / frame set) ---------------------------
with open(golden_path('item_fuzz_cases.json')) as _f:
    ITEMS = json.load(_f)


def _open_object(kind, path):
    from baseband_amd import vdif, mark5b, mark4, dada, guppi
    if kind.startswith('vdif'):
        with vdif.open(path, 'rb') as fh:
            if kind == 'vdif_frameset':
                return fh.read_frameset()
            fr = fh.read_frame()
            return fr.payload if kind == 'vdif_payload' else fr
    if kind == 'mark5b_frame':
        with mark5b.open(path, 'rb', kday=56000, nchan=8) as fh:
            return fh.read_frame()
    if kind == 'mark4_frame':
        with mark4.open(path, 'rb', ntrack=64, decade=2010) as fh:
            fh.find_header()
            return fh.read_frame()
    if kind == 'dada_frame':
        with dada.open(path, 'rb') as fh:
            return fh.read_frame(memmap=False)
    with guppi.open(path, 'rb') as fh:
        fr = fh.read_frame(memmap=False)
        return fr.payload if kind == 'guppi_payload' else fr


@pytest.mark.parametrize('k', range(len(ITEMS)), ids=['%s-%d' % (c['kind'], k) for k, c in enumerate(ITEMS)])
def test_item_access_matches_reference(k):
    """obj[item] for random items (negative ints, open / stepped slices, channel
    and thread sub-indices): shape, dtype and digest as in the reference."""
    import hashlib
    case = ITEMS[k]
    obj = _open_object(case['kind'], golden_path(case['file']))
    assert list(obj.shape) == case['shape']
    for it in case['items']:
        key = tuple(slice(v[1], v[2], v[3]) if isinstance(v, list) else v for v in it['item'])
        key = key[0] if it['bare'] else key
        if 'error' in it:
            with pytest.raises(Exception):
                obj[key]
            continue
        got = obj[key]
        d = got.cpu().numpy() if hasattr(got, 'cpu') else np.asarray(got)
        assert list(d.shape) == it['shape'], (it['item'], d.shape)
        assert str(d.dtype) == it['dtype']
        assert hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest() == it['sha256'], it['item']


def test_module_level_info_functions():
    """``<format>.info(name, **kwargs)`` as in the reference."""
    from baseband_amd import vdif, mark5b, dada
    i = vdif.info(golden_path('samples/sample.vdif'))
    assert i and i.format == 'vdif' and i.readable and tuple(i.shape) == (40000, 8)
    assert not vdif.info(golden_path('samples/sample.m5b'))
    i = mark5b.info(golden_path('samples/sample.m5b'))
    assert i.format == 'mark5b' and set(i.missing) == {'nchan', 'kday', 'ref_time'}
    i = mark5b.info(golden_path('samples/sample.m5b'), nchan=8, kday=56000)
    assert i.readable and i.used_kwargs == dict(nchan=8, kday=56000)
    assert dada.info(golden_path('samples/sample.dada')).format == 'dada'
