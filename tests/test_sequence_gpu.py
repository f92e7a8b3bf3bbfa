"""Stream readers / writers over file sequences: a
split recording must read exactly like the single file, including windows of
the staging pipeline that straddle file boundaries; a sequence writer must
produce the files the reference's writer produces
(tests/golden/sequence_cases.json)."""
import hashlib
import json
import os
import pickle

import numpy as np
import pytest

from conftest import golden_path, bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gold():
    with open(golden_path('sequence_cases.json')) as f:
        return json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def split(path_or_bytes, cuts, tmp_path, stem):
    blob = (open(path_or_bytes, 'rb').read() if isinstance(path_or_bytes, str)
            else bytes(path_or_bytes))
    edges = [0] + list(cuts) + [len(blob)]
    names = []
    for i in range(len(edges) - 1):
        names.append(str(tmp_path / ('%s%02d.bin' % (stem, i))))
        with open(names[-1], 'wb') as f:
            f.write(blob[edges[i]:edges[i + 1]])
    return names


def test_vdif_split_stream_matches_reference(gold, tmp_path):
    from baseband_amd import vdif
    g = gold['vdif_split_stream']
    names = split(golden_path('samples/sample.vdif'), gold['byte_reads']['cuts'][1:-1], tmp_path, 'v')
    with vdif.open(names, 'rs') as fs, vdif.open(golden_path('samples/sample.vdif'), 'rs') as f1:
        assert fs.shape == tuple(g['shape']) == f1.shape
        assert fs.start_time == f1.start_time and fs.stop_time == f1.stop_time
        a = fs.read().cpu().numpy()
        assert sha(a) == g['sha256']
        assert bits_equal(a, f1.read().cpu().numpy())
        fs.seek(g['seek'])
        assert sha(fs.read(g['count']).cpu().numpy()) == g['part_sha256']
        # pickling a sequence reader reopens the whole sequence
        fs.seek(777)
        with pickle.loads(pickle.dumps(fs)) as clone:
            assert clone.tell() == 777
            assert bits_equal(clone.read(50).cpu().numpy(), a[777:827])
    with vdif.open(tuple(names), 'rb') as fb:             # file reader over a sequence
        assert fb.read_header()['frame_nr'] == 0
        fb.seek(0)
        assert fb.get_thread_ids() == list(range(8))


def test_vdif_template_reader(tmp_path):
    from baseband_amd import vdif
    blob = open(golden_path('samples/sample.vdif'), 'rb').read()
    template = str(tmp_path / 'obs_{tag}.{file_nr:03d}.vdif')
    for i in range(4):                                     # 4 frames per file
        with open(template.format(tag='x1', file_nr=i), 'wb') as f:
            f.write(blob[i * 4 * 5032:(i + 1) * 4 * 5032])
    with vdif.open(template, 'rs', tag='x1') as fs, \
            vdif.open(golden_path('samples/sample.vdif'), 'rs') as f1:
        assert bits_equal(fs.read().cpu().numpy(), f1.read().cpu().numpy())
    with pytest.raises(KeyError):
        vdif.open(template, 'rs')


@pytest.mark.parametrize('fmt,sample,kwargs,cuts', [
    ('mark5b', 'samples/sample.m5b', dict(kday=56000, nchan=8, sample_rate=32e6), (10016 * 2 + 5, 33333)),
    ('mark4', 'samples/sample.m4', dict(ntrack=64, decade=2010, sample_rate=32e6), (2696 + 64 * 2500 + 9, 200001)),
    ('guppi', 'samples/sample_puppi.raw', dict(), ()),
])
def test_other_formats_split(fmt, sample, kwargs, cuts, tmp_path):
    import importlib
    mod = importlib.import_module('baseband_amd.' + fmt)
    path = golden_path(sample)
    if fmt == 'guppi':                                     # one frame per file
        with mod.open(path, 'rb') as fb:
            h = fb.read_header()
        fn = h.frame_nbytes
        cuts = tuple(range(fn, os.path.getsize(path), fn))
    names = split(path, cuts, tmp_path, fmt)
    with mod.open(names, 'rs', **kwargs) as fs, mod.open(path, 'rs', **kwargs) as f1:
        assert fs.shape == f1.shape
        assert bits_equal(fs.read().cpu().numpy(), f1.read().cpu().numpy())
        n = fs.shape[0]
        fs.seek(n // 3)
        f1.seek(n // 3)
        assert bits_equal(fs.read(n // 2).cpu().numpy(), f1.read(n // 2).cpu().numpy())


def test_windows_straddling_file_boundaries(tmp_path):
    """A multi-window read over unevenly split files: the pinned-buffer stage
    gathers each window from up to three mappings."""
    from baseband_amd import vdif, synth
    image, header0 = synth.random_vdif(11, 600, nthread=2, nchan=4, payload_nbytes=8000)
    blob = image.tobytes()
    single = str(tmp_path / 'whole.vdif')
    with open(single, 'wb') as f:
        f.write(blob)
    n = len(blob)
    cuts = [n // 7 + 13, n // 3 + 1, n // 3 + 4099, n // 2 - 8032 * 3, (3 * n) // 4 + 5]
    names = split(blob, cuts, tmp_path, 'w')
    with vdif.open(names, 'rs', sample_rate=8e6) as fs, \
            vdif.open(single, 'rs', sample_rate=8e6) as f1:
        fs.window_bytes = f1.window_bytes = 1 << 20        # ~65 frame sets per window
        a = fs.read()
        b = f1.read()
        assert a.shape == b.shape and bool((a == b).all())
        fs.seek(12345)
        f1.seek(12345)
        assert bool((fs.read(3_000_001) == f1.read(3_000_001)).all())


def test_vdif_sequence_writer_matches_reference(gold, tmp_path):
    from baseband_amd import vdif
    g = gold['vdif_sequence_write']
    with vdif.open(golden_path('samples/sample.vdif'), 'rs') as f1:
        data = f1.read()
        header0 = f1.header0
    template = str(tmp_path / 'w{file_nr:02d}.vdif')
    with vdif.open(template, 'ws', header0=header0, nthread=8, file_size=g['file_size']) as fw:
        fw.write(data)
        fw.write(data)
    files = sorted(os.listdir(str(tmp_path)))
    assert files == g['files']
    assert [os.path.getsize(str(tmp_path / f)) for f in files] == g['sizes']
    assert [hashlib.sha256(open(str(tmp_path / f), 'rb').read()).hexdigest() for f in files] == g['sha256']
    with vdif.open(template, 'rs') as fr:
        back = fr.read()
    assert bool((back == __import__('torch').cat([data, data])).all())


@pytest.mark.parametrize('fmt', ['vdif', 'dada'])
def test_sequence_writer_fills_several_files_at_once(tmp_path, fmt, monkeypatch):
    """Round 5: the writers' background sink writes the files of a sequence
    POSITIONALLY, a thread per file (`SequentialFileWriter.pwrite_stream`,
    staging._FileSink): same files, byte for byte, as the sequential path
    (BB_WRITE_ASYNC off), for file sizes that do and do not divide the pieces,
    frames that straddle files, and writes continued after a flush."""
    import torch
    import baseband_amd as bb
    from baseband_amd import staging
    from baseband_amd.vdif.header import VDIFHeader
    g = torch.Generator(device='cuda')
    g.manual_seed(5)

    def write(sub, asynchronous, file_size):
        monkeypatch.setattr(staging, '_WRITE_ASYNC', asynchronous)
        d = tmp_path / sub
        d.mkdir()
        g.manual_seed(5)
        if fmt == 'vdif':
            h0 = VDIFHeader.fromvalues(edv=0, time=np.datetime64('2014-06-13T05:30:01'), nchan=1, bps=2,
                                       complex_data=False, thread_id=0, samples_per_frame=32000, station='AA')
            fw = bb.vdif.open(str(d / 'f{file_nr:03d}.vdif'), 'ws', header0=h0, sample_rate=32e6, nthread=1,
                              file_size=file_size)
            chunks = [torch.randn(n * 32000, device='cuda', generator=g) * 2. for n in (700, 1, 2300, 64)]
        else:
            from baseband_amd.dada.header import DADAHeader
            h0 = DADAHeader.fromvalues(time=np.datetime64('2013-07-02T01:39:20'), offset=0., sample_rate=16e6, bps=8,
                                       complex_data=True, npol=2, nchan=1, payload_nbytes=4 << 20,
                                       start_time=np.datetime64('2013-07-02T01:39:20'), telescope='GMRT')
            fw = bb.dada.open(str(d / '{utc_start}_{obs_offset:016d}.{file_nr:06d}.dada'), 'ws', header0=h0)
            spf = h0.samples_per_frame
            chunks = [(torch.randn(n, 2, 2, device='cuda', generator=g) * 20.) for n in (3 * spf, spf // 2, spf // 2, 5 * spf)]
            chunks = [torch.view_as_complex(c.contiguous()) for c in chunks]
        with fw:
            for k, c in enumerate(chunks):
                fw.write(c)
                if k == 1:
                    fw.flush()
        return {n: hashlib.sha256(open(d / n, 'rb').read()).hexdigest() for n in sorted(os.listdir(d))}

    sizes = [8032 * 400, 8032 * 1000 + 4000, 7_000_000] if fmt == 'vdif' else [None]
    for fs in sizes:
        tag = str(fs)
        want = write('seq_' + tag, False, fs)
        got = write('par_' + tag, True, fs)
        assert len(want) >= 3 and got == want, (fs, sorted(want), sorted(got))


def test_background_sink_writes_the_same_single_file(tmp_path, monkeypatch):
    """One file, 330 MB, written in calls of very different sizes with temporaries
    coming and going in between: the background sink (pieces in flight across
    write() calls) gives the file the synchronous path gives."""
    import torch
    import baseband_amd as bb
    from baseband_amd import staging
    from baseband_amd.vdif.header import VDIFHeader
    h0 = VDIFHeader.fromvalues(edv=0, time=np.datetime64('2014-06-13T05:30:01'), nchan=1, bps=2,
                               complex_data=False, thread_id=0, samples_per_frame=32000, station='AA')
    g = torch.Generator(device='cuda')

    def write(name, asynchronous):
        monkeypatch.setattr(staging, '_WRITE_ASYNC', asynchronous)
        g.manual_seed(11)
        path = tmp_path / name
        with bb.vdif.open(str(path), 'ws', header0=h0, sample_rate=32e6, nthread=1) as fw:
            for n in (9000, 1, 17, 12000, 300, 8000, 5, 11000, 640):
                x = torch.randn(n * 32000, device='cuda', generator=g) * 2.
                fw.write(x)
                junk = (x[:1000].double() * 3).sum()        # temporaries between the calls
                del x, junk
        return hashlib.sha256(path.read_bytes()).hexdigest(), path.stat().st_size

    want = write('sync.vdif', False)
    got = write('async.vdif', True)
    assert got == want and want[1] == 40963 * 8032
