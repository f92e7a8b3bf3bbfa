"""Decoded read-ahead of small sequential reads (`_from_decoded_ahead`): the
frame-at-a-time loop must give exactly what exact per-read decoding gives --
samples, errors and the read at which they occur."""
import json
import warnings

import numpy as np
import pytest

from conftest import golden_path, load_expected, load_file, bits_equal

pytestmark = pytest.mark.gpu


def _open(name, manifest, **kw):
    import baseband_amd
    from baseband_amd import vdif, mark5b, mark4, gsb
    case = manifest[name]
    if name.startswith('sample_gsb'):
        if 'files' in case:
            raw = [[golden_path(f) for f in pol] for pol in case['files']]
            return gsb.open(golden_path(case['timestamp']), 'rs', raw=raw, samples_per_frame=8, **kw)
        return gsb.open(golden_path(case['timestamp']), 'rs', raw=golden_path(case['file']),
                        samples_per_frame=8192, **kw)
    path = golden_path(case['file'])
    if 'vdif' in name:
        if 'frame_rate' in case:
            kw.setdefault('sample_rate', case['frame_rate'] * case['samples_per_frame'])
        return vdif.open(path, 'rs', **kw)
    if name.startswith('m5b') or 'mark5b' in name or name == 'sample_m5b':
        return mark5b.open(path, 'rs', nchan=case.get('nchan', 8), bps=case.get('bps', 2), kday=56000,
                           sample_rate=case['frame_rate'] * case['samples_per_frame'], **kw)
    return mark4.open(path, 'rs', ntrack=case['ntrack'], decade=2010,
                      sample_rate=case['frame_rate'] * case['samples_per_frame'], **kw)


CASES = ['sample_vdif', 'vdif_cfg2_small', 'vdif_cfg3_small', 'm5b_c16_b2', 'm4_t64_f4',
         'sample_gsb_rawdump', 'sample_gsb_phased']


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('step', ['frame', 'odd', 'tiny'])
def test_sequential_small_reads_equal_one_read(manifest, name, step):
    exp = load_expected(name)
    with _open(name, manifest, squeeze=False) as fh:
        n, spf = fh.shape[0], fh.samples_per_frame
        chunk = {'frame': spf, 'odd': max(1, spf // 3 + 7), 'tiny': max(1, min(spf, 5))}[step]
        nread = min(-(-n // chunk), 400)
        used, pos = 0, 0
        for _ in range(nread):
            cnt = min(chunk, n - pos)
            got = fh.read(cnt).cpu().numpy()
            assert bits_equal(got, np.ascontiguousarray(exp[pos:pos + cnt])), (name, pos, cnt)
            pos += cnt
            used += fh._decoded is not None
            assert fh.tell() == pos
        if nread > 4:
            assert used > 0, "the loop never ran from the decoded window"
        # a seek drops the window; reads stay right, and a new sequence builds a new one
        for off in (1, max(0, n // 2 - 3), 0):
            fh.seek(off)
            cnt = min(chunk, n - off)
            assert bits_equal(fh.read(cnt).cpu().numpy(), np.ascontiguousarray(exp[off:off + cnt]))
            assert fh._decoded is None


def test_window_samples_are_served_once_and_out_is_filled(manifest):
    import torch
    exp = load_expected('vdif_cfg2_small')
    with _open('vdif_cfg2_small', manifest) as fh:
        spf = fh.samples_per_frame
        want = exp.reshape(exp.shape[0], -1)[:, 0] if exp.ndim > 1 else exp
        pos = 0
        first = []
        for k in range(9):
            if k % 3 == 0:
                out = torch.empty((spf,) + fh.sample_shape, dtype=torch.float32, device='cuda')
                got = fh.read(out=out)
                assert got is out
            elif k % 3 == 1:
                host = np.empty((spf,) + fh.sample_shape, np.float32)
                got = torch.from_numpy(fh.read(out=host))
            else:
                got = fh.read(spf)
                got.mul_(2.)                    # in-place use of a served view ...
                got = got / 2.
            first.append(got.cpu().numpy())
            pos += spf
        assert bits_equal(np.concatenate(first).reshape(-1), np.ascontiguousarray(want[:pos]).reshape(-1))
        # ... is never seen again: going back decodes afresh
        fh.seek(2 * spf)
        assert bits_equal(fh.read(spf).cpu().numpy().reshape(-1), np.ascontiguousarray(want[2 * spf:3 * spf]).reshape(-1))


@pytest.mark.parametrize('verify', [True, 'fix'])
def test_problems_surface_at_the_same_read_as_without_read_ahead(tmp_path, verify):
    """A file with a damaged frame further on: the loop with read-ahead raises
    (verify=True) or warns and repairs (verify='fix') at the same read, with
    the same samples, as exact per-read decoding."""
    from baseband_amd import vdif
    base = load_file('synth/vdif_triple.bin').copy()
    with vdif.open(golden_path('synth/vdif_triple.bin'), 'rs') as fh:
        fnb = fh.header0.frame_nbytes
        nthread = fh._unsliced_shape[0] if len(fh._unsliced_shape) > 1 else 1
        spf, n = fh.samples_per_frame, fh.shape[0]
    nsets = n // spf
    bad_set = min(nsets - 2, 40)
    blob = np.delete(base, np.arange(bad_set * nthread * fnb + 100, bad_set * nthread * fnb + 100 + fnb))
    p = tmp_path / 'damaged.vdif'
    p.write_bytes(blob.tobytes())
    chunk = max(1, spf // 4)

    def loop(ahead):
        events, data = [], []
        with vdif.open(str(p), 'rs', verify=verify) as fh:
            fh.decode_ahead = ahead
            for k in range(min(400, fh.shape[0] // chunk)):
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter('always')
                    try:
                        data.append(fh.read(chunk).cpu().numpy())
                    except (ValueError, EOFError, OSError, AssertionError) as exc:
                        # (which of them: the first problem the reference's set-by-set loop meets,
                        # tests/golden/refcases/damaged_streams.json)
                        events.append((k, type(exc).__name__))
                        break
                if w:
                    events.append((k, 'warning'))
        return events, np.concatenate(data) if data else np.empty(0)

    ev_exact, d_exact = loop(False)
    ev_ahead, d_ahead = loop(True)
    assert ev_exact and ev_exact == ev_ahead, (ev_exact, ev_ahead)
    assert bits_equal(d_exact, d_ahead)


@pytest.mark.parametrize('fmt,name', [('guppi', 'guppi_cf_c64_ov0'), ('guppi', 'guppi_cf_c64_ov32'),
                                      ('guppi', 'guppi_tf_c8_ov16'), ('dada', 'dada_p2_c4_cplx'),
                                      ('dada', 'sample_meerkat_dada')])
def test_block_formats_prefetch_the_next_frame(manifest, fmt, name):
    """Loops of small reads on block formats: once a frame is staged whole, the
    next one travels to HBM on the worker thread; the samples are those of the
    one-shot read (pinned to the reference), also after seeks that leave a
    prefetch unused."""
    import baseband_amd
    from baseband_amd import staging
    mod = getattr(baseband_amd, fmt)
    case = manifest[name]
    calls = []
    real = staging.upload_in_background

    def counting(*a, **k):
        calls.append(a[1:3])
        return real(*a, **k)
    staging.upload_in_background = counting
    try:
        with mod.open(golden_path(case['file']), 'rs', squeeze=False) as fh:
            whole = fh.read().cpu().numpy()
            n, spf = fh.shape[0], fh.samples_per_frame
            chunk = max(1, spf // 5 + 1)
            for start in (0, 3):
                fh.seek(start)
                pos = start
                while pos < n:
                    cnt = min(chunk, n - pos)
                    # (a read that starts inside a later GUPPI block takes that block's own
                    # overlap rows: compare with the same read done afresh)
                    with mod.open(golden_path(case['file']), 'rs', squeeze=False) as ref:
                        ref.seek(pos)
                        want = ref.read(cnt).cpu().numpy()
                    assert bits_equal(fh.read(cnt).cpu().numpy(), want), (name, pos, cnt)
                    pos += cnt
            if fh._nframes > 1:
                assert calls, "no frame was prefetched"
            # leave a prefetch behind and go elsewhere
            fh.seek(0)
            fh.read(chunk)
            fh.read(chunk)
            fh.seek(n - 2)
            assert bits_equal(fh.read(2).cpu().numpy(), whole[n - 2:])
    finally:
        staging.upload_in_background = real


@pytest.mark.parametrize('name', ['sample_vdif', 'vdif_cfg2_small', 'vdif_cfg3_small', 'm5b_c16_b2', 'm4_t64_f4', 'm4_t16_f4'])
def test_frame_loops_decode_from_a_window_in_hbm(manifest, name):
    """read_frame().data in a loop ('rb'): from the third frame on the payload
    decodes from a window of the file kept in HBM (`FileBase._lend_device_words`);
    every frame equals the frame read on its own."""
    from baseband_amd import vdif, mark5b, mark4
    case = manifest[name]
    path = golden_path(case['file'])
    if name.startswith('m4'):
        opener = lambda: mark4.open(path, 'rb', ntrack=case['ntrack'], decade=2010)     # noqa: E731
    elif name.startswith('m5b'):
        opener = lambda: mark5b.open(path, 'rb', nchan=case['nchan'], bps=case['bps'], kday=56000)   # noqa: E731
    else:
        opener = lambda: vdif.open(path, 'rb')       # noqa: E731
    with opener() as fh:
        fh.seek(0, 2)
        size = fh.tell()
        fh.seek(0)
        lent, k = 0, 0
        while fh.tell() < size and k < 200:
            pos = fh.tell()
            frame = fh.read_frame()
            lent += frame.payload._dwords is not None
            got = frame.data.cpu().numpy()
            with opener() as ref:                   # the same frame, read alone
                ref.seek(pos)
                alone = ref.read_frame()
                assert alone.payload._dwords is None
                assert bits_equal(got, alone.data.cpu().numpy()), (name, k)
            k += 1
        assert k < 4 or lent >= k - 3
        # a seek breaks the sequence: the next frames are on their own again
        fh.seek(0)
        assert fh.read_frame().payload._dwords is None
        assert fh.read_frame().payload._dwords is None
        frame = fh.read_frame()
        if frame.payload._dwords is not None:
            # (bytes read from a file are read-only, here as in the reference:
            # the borrowed device bytes cannot go stale)
            with pytest.raises(ValueError):
                frame.payload[0] = frame.payload[1]


@pytest.mark.parametrize('name,subset', [('sample_vdif', ([1, 3],)), ('vdif_cfg3_small', (slice(None), [3, 9])),
                                         ('vdif_cfg3_small', ([6, 1], slice(2, 7))), ('m5b_c16_b2', ([1, 6, 7],)),
                                         ('m4_t64_f4', ([5, 0],)), ('sample_gsb_phased', (slice(None), slice(10, 20)))])
def test_sequential_small_reads_with_subsets(manifest, name, subset):
    """The decoded window holds what read() returns -- after squeeze and subset
    (thread subsets through the index, channel subsets folded into the decode)."""
    exp = load_expected(name)
    with _open(name, manifest, subset=subset) as fh:
        want = exp.reshape((exp.shape[0],) + tuple(s for s in exp.shape[1:] if s > 1))[(slice(None),) + subset]
        assert fh.shape == want.shape
        n, spf = fh.shape[0], fh.samples_per_frame
        chunk = max(1, spf // 2 + 3)
        pos, used = 0, 0
        while pos < n and pos < 60 * chunk:
            cnt = min(chunk, n - pos)
            assert bits_equal(fh.read(cnt).cpu().numpy(), np.ascontiguousarray(want[pos:pos + cnt])), (name, pos)
            pos += cnt
            used += fh._decoded is not None
        assert used > 0 or n <= 3 * chunk



FUZZ_CASES = CASES + ['guppi_cf_c64_ov0', 'guppi_cf_c64_ov32', 'guppi_tf_c8_ov16', 'dada_p2_c4_cplx', 'sample_meerkat_dada']


def _open_any(name, manifest, **kw):
    import baseband_amd
    if name.startswith(('guppi', 'sample_puppi')):
        return baseband_amd.guppi.open(golden_path(manifest[name]['file']), 'rs', **kw)
    if name.startswith('dada') or name.endswith('_dada'):
        return baseband_amd.dada.open(golden_path(manifest[name]['file']), 'rs', **kw)
    return _open(name, manifest, **kw)


@pytest.mark.parametrize('name', FUZZ_CASES)
@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_random_walks_equal_exact_reads(manifest, name, seed):
    """Seeded random walks over a stream -- runs of sequential reads of all sizes,
    seeks, reads into device and host `out` -- with every read-ahead on, against
    a reader that decodes exactly what each read asks for (which the golden
    stream tests pin to the reference)."""
    import torch
    rng = np.random.default_rng(1000 * seed + len(name))
    with _open_any(name, manifest, squeeze=False) as fh, _open_any(name, manifest, squeeze=False) as ref:
        ref.decode_ahead = False
        if hasattr(ref, 'prefetch_next'):
            ref.prefetch_next = False
        n, spf = fh.shape[0], fh.samples_per_frame
        pos = 0
        for step in range(150):
            kind = rng.integers(10)
            if kind == 0 or pos >= n:                      # seek somewhere
                pos = int(rng.integers(0, n))
                fh.seek(pos)
            size = int(rng.choice([1, 3, max(1, spf // 7), spf, spf + 5, 3 * spf + 1, max(1, n // 3)]))
            cnt = max(1, min(size, n - pos))
            ref.seek(pos)
            want = ref.read(cnt).cpu().numpy()
            how = rng.integers(4)
            if how == 0:
                out = torch.empty((cnt,) + fh.sample_shape, dtype=torch.complex64 if fh.complex_data else torch.float32,
                                  device='cuda')
                got = fh.read(out=out).cpu().numpy()
            elif how == 1:
                host = np.empty((cnt,) + fh.sample_shape, np.complex64 if fh.complex_data else np.float32)
                got = fh.read(out=host)
            else:
                got = fh.read(cnt).cpu().numpy()
            assert bits_equal(got, want), (name, seed, step, pos, cnt, int(how))
            pos += cnt
            assert fh.tell() == pos


@pytest.mark.parametrize('name,subset', [('sample_vdif', ([5, 2],)), ('vdif_cfg3_small', (slice(None), [9, 3])),
                                         ('vdif_cfg3_small', ([0, 7], slice(4, 12))), ('m5b_c16_b2', ([3, 12],)),
                                         ('m4_t64_f4', (slice(2, 7),)), ('sample_gsb_phased', (slice(None), [511, 0])),
                                         ('guppi_cf_c64_ov32', (slice(None), slice(8, 40))),
                                         ('guppi_tf_c8_ov16', (slice(None), slice(2, 6))),
                                         ('dada_p2_c4_cplx', (slice(None), [3, 1])), ('sample_meerkat_dada', (0,))])
def test_random_walks_with_subsets(manifest, name, subset):
    """The same walks with reader subsets (thread subsets, folded channel
    selections, channel ranges) and squeezing."""
    rng = np.random.default_rng(len(name) + len(repr(subset)))
    with _open_any(name, manifest, subset=subset) as fh, _open_any(name, manifest, subset=subset) as ref:
        ref.decode_ahead = False
        if hasattr(ref, 'prefetch_next'):
            ref.prefetch_next = False
        assert fh.shape == ref.shape
        n, spf = fh.shape[0], fh.samples_per_frame
        pos = 0
        for step in range(120):
            if rng.integers(8) == 0 or pos >= n:
                pos = int(rng.integers(0, n))
                fh.seek(pos)
            cnt = max(1, min(int(rng.choice([1, 4, max(1, spf // 5), spf, 2 * spf + 3])), n - pos))
            ref.seek(pos)
            assert bits_equal(fh.read(cnt).cpu().numpy(), ref.read(cnt).cpu().numpy()), (name, subset, step, pos, cnt)
            pos += cnt


def test_random_walk_over_a_damaged_file(tmp_path):
    """verify='fix' on a file with bytes missing: walks with and without
    read-ahead return the same (repaired) samples."""
    from baseband_amd import vdif
    base = load_file('synth/vdif_triple.bin').copy()
    blob = np.delete(base, np.arange(3 * 8 * 5032 + 77, 3 * 8 * 5032 + 77 + 5032))
    p = tmp_path / 'damaged.vdif'
    p.write_bytes(blob.tobytes())
    rng = np.random.default_rng(5)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with vdif.open(str(p), 'rs') as fh, vdif.open(str(p), 'rs') as ref:
            ref.decode_ahead = False
            n, spf = fh.shape[0], fh.samples_per_frame
            pos = 0
            for step in range(200):
                if rng.integers(6) == 0 or pos >= n:
                    pos = int(rng.integers(0, n))
                    fh.seek(pos)
                cnt = max(1, min(int(rng.choice([1, spf // 9, spf // 2, spf, 2 * spf + 1])), n - pos))
                ref.seek(pos)
                assert bits_equal(fh.read(cnt).cpu().numpy(), ref.read(cnt).cpu().numpy()), (step, pos, cnt)
                pos += cnt


def test_small_results_are_copies_and_large_ones_views(manifest):
    """A read served from the decoded window returns a fresh tensor when it is
    small (the reference returns fresh arrays, base/base.py:919-969; a view
    would pin the whole window) and a view when copying would cost a second
    pass (`decode_ahead_copy_below`)."""
    with _open('vdif_cfg2_small', manifest) as fh:
        spf = fh.samples_per_frame
        for _ in range(4):
            fh.read(spf)
        assert fh._decoded is not None
        win = fh._decoded[2]
        lo, hi = win.data_ptr(), win.data_ptr() + win.numel() * win.element_size()
        small = fh.read(spf)
        assert small.numel() * small.element_size() < fh.decode_ahead_copy_below
        assert not (lo <= small.data_ptr() < hi), "small read-ahead result aliases the window"
        assert small.untyped_storage().nbytes() == small.numel() * small.element_size()
        fh.decode_ahead_copy_below = 0
        view = fh.read(spf)
        assert fh._decoded is not None and lo <= view.data_ptr() < hi


def test_read_ahead_failures_that_are_not_about_the_file_surface(manifest, monkeypatch):
    """Out of memory for a speculative window: a warning, read-ahead off, the
    read itself still served.  Anything else (library errors, bugs) is not
    swallowed."""
    import torch
    from baseband_amd.vdif.base import VDIFStreamReader
    exp = load_expected('vdif_cfg2_small')
    real = VDIFStreamReader._read_sets              # (before anything is patched)
    with _open('vdif_cfg2_small', manifest, squeeze=False) as fh:
        spf = fh.samples_per_frame
        state = {'n': 0}

        def oom_once(self, first, last, into=None):
            if last - first > 4 and state['n'] == 0:
                state['n'] = 1
                raise torch.cuda.OutOfMemoryError("simulated")
            return real(self, first, last, into)
        monkeypatch.setattr(type(fh), '_read_sets', oom_once)
        pos = 0
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            for _ in range(5):
                got = fh.read(spf).cpu().numpy()
                assert bits_equal(got, np.ascontiguousarray(exp[pos:pos + spf]))
                pos += spf
        assert state['n'] == 1 and fh.decode_ahead is False
        assert any('read-ahead switched off' in str(x.message) for x in w)
    with _open('vdif_cfg2_small', manifest, squeeze=False) as fh:
        spf = fh.samples_per_frame

        def broken(self, first, last, into=None):
            if last - first > 4:
                raise RuntimeError("simulated library failure")
            return real(self, first, last, into)
        monkeypatch.setattr(type(fh), '_read_sets', broken)
        with pytest.raises(RuntimeError, match="simulated library failure"):
            for _ in range(5):
                fh.read(spf)


def test_abandoned_windows_leave_no_bad_frame_count_behind(manifest):
    """`_reset_checks` clears the device counter too: counts from windows of
    a read that did not complete must not be blamed on the next read."""
    import torch
    with _open('vdif_cfg2_small', manifest) as fh:
        fh.read(fh.samples_per_frame)
        assert fh._nbad is not None
        fh._nbad.fill_(3)
        fh._nmissing, fh._checked = 2, True
        fh._reset_checks()
        assert int(fh._nbad.item()) == 0 and fh._nmissing == 0 and fh._checked is False
        fh.seek(0)
        fh.read(fh.samples_per_frame)            # no spurious "wrong frame number"


def test_host_results_small_read_loops_come_from_the_window_mirror(tmp_path):
    """``host_results = True`` (what the plugin modules switch on): ``read()`` returns
    new NumPy arrays; in a loop of small sequential reads they are cut out of ONE host
    copy per decoded window.  Same samples as the device path, read for read, also
    across window refills, a seek, and with verify off."""
    import numpy as np
    import torch
    from baseband_amd import vdif, synth
    image, h0 = synth.random_vdif(11, 700, payload_nbytes=8000, frame_rate=1000)
    p = tmp_path / 'loop.vdif'
    image.tofile(str(p))
    with vdif.open(str(p), 'rs', sample_rate=32e6) as ref:
        want = ref.read().cpu().numpy()
    for verify in ('fix', False):
        with vdif.open(str(p), 'rs', sample_rate=32e6, verify=verify) as fh:
            fh.host_results = True
            pos, sizes, mirrors = 0, [32000] * 40 + [5000, 27000, 64000, 1000] * 20 + [32000] * 100, set()
            for n in sizes:
                got = fh.read(n)
                assert isinstance(got, np.ndarray) and got.flags.writeable and got.shape == (n,)
                assert np.array_equal(got, want[pos:pos + n]), pos
                pos += n
                if fh._decoded_host is not None:
                    mirrors.add(id(fh._decoded_host[1]))
            assert 1 <= len(mirrors) <= 8                   # a few windows, not one copy per read
            got[:] = 0                                      # results are the caller's own
            fh.seek(123)
            assert np.array_equal(fh.read(1000), want[123:1123])
            big = fh.read(3000000)                          # a large result: its own pinned array
            assert isinstance(big, np.ndarray) and np.array_equal(big, want[1123:3001123])
            out = torch.empty(32000, device='cuda')
            assert fh.read(out=out) is out                  # `out` decides, as before
            fh.host_results = False
            assert isinstance(fh.read(32000), torch.Tensor)
