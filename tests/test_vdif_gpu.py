"""VDIF through the drop-in API (open/read/seek, Payload, Frame, FrameSet)
on the GPU, bit-exact against the reference's golden outputs and the oracle."""
import io

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import golden_path, load_expected, load_file, bits_equal

pytestmark = pytest.mark.gpu

ALL = ['sample_vdif', 'sample_mwa_vdif', 'sample_arochime_vdif',
       'sample_bps1_vdif', 'vdif_cfg2_small', 'vdif_cfg3_small',
       'vdif_bps1_c4', 'vdif_bps4_cplx_t2', 'vdif_bps8_real_c2',
       'vdif_bps8_cplx_t4', 'vdif_bps2_t8_c1', 'vdif_legacy_bps2',
       'vdif_bps4_t2_c1', 'vdif_invalid_fill0', 'vdif_invalid_fillm999']


def _kw(case):
    kw = {}
    if 'frame_rate' in case:
        kw['sample_rate'] = case['frame_rate'] * case['samples_per_frame']
    elif case.get('kwargs'):
        kw['sample_rate'] = case['kwargs']['sample_rate']
    if 'fill_value' in case:
        kw['fill_value'] = case['fill_value']
    return kw


@pytest.mark.parametrize('name', ALL)
def test_stream_read_matches_reference(manifest, name):
    from baseband_amd import vdif
    case = manifest[name]
    with vdif.open(golden_path(case['file']), 'rs', squeeze=False, **_kw(case)) as fh:
        assert fh.shape == tuple(case['shape'])
        data = fh.read()
        assert fh.tell() == fh.shape[0]
        with pytest.raises(EOFError):
            fh.read(1)
    assert data.is_cuda
    assert bits_equal(data.cpu().numpy(), load_expected(name))


def test_stream_read_from_memory_file(manifest):
    from baseband_amd import vdif
    case = manifest['vdif_cfg3_small']
    blob = load_file(case['file']).tobytes()
    with vdif.open(io.BytesIO(blob), 'rs', squeeze=False, **_kw(case)) as fh:
        assert bits_equal(fh.read().cpu().numpy(), load_expected('vdif_cfg3_small'))


@pytest.mark.parametrize('name', ['sample_vdif', 'vdif_cfg3_small', 'vdif_bps4_cplx_t2'])
def test_partial_reads_and_seek(manifest, name):
    """Reads that start/stop inside frames and cross frame boundaries
    (vdif/tests/test_vdif.py:876-951)."""
    from baseband_amd import vdif
    case = manifest[name]
    exp = load_expected(name)
    spf = case['samples_per_frame']
    n = exp.shape[0]
    with vdif.open(golden_path(case['file']), 'rs', squeeze=False, **_kw(case)) as fh:
        first = fh.read(12)
        assert bits_equal(first.cpu().numpy(), exp[:12])
        assert fh.tell() == 12
        for start, count in ((spf - 3, 7), (spf, spf), (1, spf + 5),
                             (n - 5, 5), (spf // 2, 1), (0, 0)):
            fh.seek(start)
            out = fh.read(count)
            assert bits_equal(out.cpu().numpy(), exp[start:start + count]), (start, count)
            assert fh.tell() == start + count
        fh.seek(-10, 2)
        assert bits_equal(fh.read().cpu().numpy(), exp[-10:])
        fh.seek(n - 3)
        with pytest.raises(EOFError):
            fh.read(4)
        # out= argument: device tensor and host array
        import torch
        fh.seek(5)
        out = torch.empty((9,) + exp.shape[1:], dtype=first.dtype, device='cuda')
        assert fh.read(out=out) is out
        assert bits_equal(out.cpu().numpy(), exp[5:14])
        fh.seek(5)
        hout = np.empty((9,) + exp.shape[1:], exp.dtype)
        fh.read(out=hout)
        assert bits_equal(hout, exp[5:14])
        with pytest.raises(AssertionError):
            fh.read(out=np.empty((3, 99), exp.dtype))
    with pytest.raises(ValueError):
        fh.read(1)                                   # closed


def test_sample_vdif_known_answers(manifest):
    """vdif/tests/test_vdif.py:930-931: stream read(12)[:, 0]."""
    from baseband_amd import vdif
    with vdif.open(golden_path('samples/sample.vdif'), 'rs') as fh:
        d = fh.read(12).cpu().numpy()
    assert d.shape == (12, 8)
    assert d[:, 0].astype(int).tolist() == [-1, -1, 3, -1, 1, -1, 3, -1, 1, 3, -1, 1]
    assert d[:, 3].astype(int).tolist() == [-1, 1, -1, 1, -3, -1, 3, -1, 3, -3, 1, 3]


@pytest.mark.parametrize('squeeze,subset', [
    (True, ()), (False, ()), (True, [1, 3]), (True, 2), (False, 2),
    (False, (slice(1, 7, 2), 0)), (True, slice(4, None)), (True, [5]),
    (False, ([0, 7],)), (False, (slice(None), slice(None)))])
def test_squeeze_and_subset(manifest, squeeze, subset):
    """Subset/squeeze matrix (vdif/tests/test_vdif.py:1072-1149): equals
    NumPy indexing of the full reference output."""
    from baseband_amd import vdif
    exp = load_expected('sample_vdif')           # (40000, 8, 1)
    full = exp.reshape(40000, 8) if squeeze else exp
    sub = subset if isinstance(subset, tuple) else (subset,)
    want = full[(slice(None),) + sub] if sub else full
    with vdif.open(golden_path('samples/sample.vdif'), 'rs', squeeze=squeeze,
                   subset=subset) as fh:
        assert fh.sample_shape == want.shape[1:]
        fh.seek(19990)
        got = fh.read(30).cpu().numpy()
    assert bits_equal(got, np.ascontiguousarray(want[19990:20020]))


def test_subset_channels_multichannel(manifest):
    from baseband_amd import vdif
    case = manifest['vdif_cfg3_small']
    exp = load_expected('vdif_cfg3_small')       # (4000, 8, 16) c64
    with vdif.open(golden_path(case['file']), 'rs',
                   subset=([6, 1], slice(3, 9)), **_kw(case)) as fh:
        got = fh.read().cpu().numpy()
    assert bits_equal(got, np.ascontiguousarray(exp[:, [6, 1], 3:9]))


def test_verify_modes_on_corrupt_header(manifest, tmp_path):
    from baseband_amd import vdif
    case = manifest['vdif_bps2_t8_c1']
    blob = bytearray(load_file(case['file']).tobytes())
    blob[5032 * 3 + 21] ^= 0x55                  # break sync pattern of frame 3
    p = tmp_path / 'corrupt.vdif'
    p.write_bytes(bytes(blob))
    exp = load_expected('vdif_bps2_t8_c1').copy()
    with vdif.open(str(p), 'rs', squeeze=False, verify=True, **_kw(case)) as fh:
        # (the reference's header verification: tests/golden/refcases/damaged_streams.json,
        # vdif_headers_damaged_in_place)
        with pytest.raises(AssertionError):
            fh.read()
    with vdif.open(str(p), 'rs', squeeze=False, verify='fix', **_kw(case)) as fh:
        with pytest.warns(UserWarning):
            got = fh.read().cpu().numpy()
    bad_thread = case['thread_ids'].index(7)      # 4th stored frame is thread 7
    exp[:20000, bad_thread] = 0.
    assert bits_equal(got, exp)                  # only the damaged frame is lost
    with vdif.open(str(p), 'rs', squeeze=False, verify=False, **_kw(case)) as fh:
        fh.read()                                # no check, no error


PAYLOAD_ITEMS = [(), 0, -1, slice(None), slice(2, 11), slice(5, 6),
                 slice(1, None, 3), (slice(3, 29), 0), (7, slice(None)),
                 slice(-9, None), (slice(None, None, 2), -1)]


@pytest.mark.parametrize('name', ['vdif_cfg3_small', 'vdif_bps1_c4',
                                  'vdif_bps8_real_c2', 'vdif_bps4_t2_c1'])
@pytest.mark.parametrize('item', PAYLOAD_ITEMS)
def test_payload_fromfile_and_getitem(manifest, name, item):
    """Payload.fromfile / .data / __getitem__ (base/payload.py:84-139,327-330;
    item matrix of vdif/tests/test_vdif.py:431-449)."""
    from baseband_amd.vdif import VDIFHeader, VDIFPayload
    case = manifest[name]
    with open(golden_path(case['file']), 'rb') as f:
        header = VDIFHeader.fromfile(f)
        pl = VDIFPayload.fromfile(f, header)
        raw = np.frombuffer(pl.words.tobytes(), np.uint8)
    full = orc.payload_data(raw, 'vdif', header.bps, (header.nchan,),
                            header.complex_data)
    assert pl.shape == full.shape and pl.dtype == full.dtype
    assert len(pl) == header.samples_per_frame
    want = full[item]
    got = pl[item]
    got = got.cpu().numpy()
    assert got.shape == np.shape(want)
    assert bits_equal(got.reshape(-1), np.ascontiguousarray(want).reshape(-1))
    if item == ():
        assert bits_equal(np.asarray(pl), full)
        assert bits_equal(pl.data.cpu().numpy(), full)


def test_payload_unsupported_bps_is_keyerror():
    """A coder the reference has no decoder for surfaces as KeyError
    (base/payload.py:314-315; vdif/tests/test_vdif.py:414-429)."""
    from baseband_amd.vdif import VDIFPayload
    pl = VDIFPayload(np.zeros(16, '<u4'), sample_shape=(1,), bps=7)
    with pytest.raises(KeyError):
        pl.data


def test_frame_and_frameset(manifest):
    from baseband_amd import vdif
    exp = load_expected('sample_vdif')
    with vdif.open(golden_path('samples/sample.vdif'), 'rb') as fb:
        frame = fb.read_frame()
        assert frame.header['thread_id'] == 1 and frame.valid
        assert frame.shape == (20000, 1)
        d = frame.data.cpu().numpy()
        assert bits_equal(d, exp[:20000, 1])
        # vdif/tests/test_vdif.py:381-382
        assert d[:12, 0].astype(int).tolist() == [1, 1, 1, -3, 1, 1, -3, -3, -3, 3, 3, -1]
        assert bits_equal(frame[5:9].cpu().numpy(), exp[5:9, 1])
        frame.valid = False
        assert np.all(frame.data.cpu().numpy() == 0.)
        frame.fill_value = 3.5
        assert np.all(frame[10:20].cpu().numpy() == 3.5)
        fb.seek(0)
        fs = fb.read_frameset()
        assert fs.shape == (20000, 8, 1)
        assert bits_equal(fs.data.cpu().numpy(), exp[:20000])
        assert fs['thread_id'].tolist() == list(range(8))
        fs2 = fb.read_frameset(thread_ids=[3, 6])
        assert bits_equal(fs2.data.cpu().numpy(), exp[20000:, [3, 6]])
        assert bits_equal(fs2[100:110, 1].cpu().numpy(), exp[20100:20110, 6])


def test_frameset_with_invalid_frames(manifest):
    from baseband_amd import vdif
    case = manifest['vdif_invalid_fillm999']
    exp = load_expected('vdif_invalid_fillm999')
    spf = case['samples_per_frame']
    with vdif.open(golden_path(case['file']), 'rb') as fb:
        for i in range(case['nframes']):
            fs = fb.read_frameset()
            fs.fill_value = -999.
            assert bits_equal(fs.data.cpu().numpy(), exp[i * spf:(i + 1) * spf])


def test_top_level_open(manifest):
    import baseband_amd
    with baseband_amd.open(golden_path('samples/sample.vdif'), 'rs', format='vdif') as fh:
        assert fh.shape == (40000, 8)
    with baseband_amd.open(golden_path('samples/sample.vdif'), 'rs') as fh:    # auto-detected
        assert fh.shape == (40000, 8)
    with pytest.raises(ValueError):
        baseband_amd.open(golden_path('samples/sample.vdif'), 'rs', format='nonsense')
    with pytest.raises(ValueError):
        baseband_amd.vdif.open(golden_path('samples/sample.vdif'), 'xs')
    # an invalid writer call must not touch an existing file
    import os
    size = os.path.getsize(golden_path('samples/sample.vdif'))
    with pytest.raises(ValueError):
        baseband_amd.vdif.open(golden_path('samples/sample.vdif'), 'ws', edv=0, bps=2,
                               nchan=1, samples_per_frame=32000)        # no sample rate
    assert os.path.getsize(golden_path('samples/sample.vdif')) == size


@pytest.mark.parametrize('cfg', [
    dict(nthread=1, nchan=1, bps=2, complex_data=False, payload_nbytes=8000, nsets=700),
    dict(nthread=8, nchan=16, bps=2, complex_data=True, payload_nbytes=8000, nsets=90),
    dict(nthread=8, nchan=1, bps=2, complex_data=False, edv=3, nsets=200),
    dict(nthread=2, nchan=4, bps=4, complex_data=True, payload_nbytes=4096, nsets=300),
    dict(nthread=4, nchan=2, bps=8, complex_data=False, payload_nbytes=1024, nsets=500),
    dict(nthread=3, nchan=8, bps=1, complex_data=False, payload_nbytes=2048, nsets=257),
])
def test_seeded_synthetic_vs_oracle(cfg, tmp_path):
    """Medium files (several MiB, multiple pipeline windows) vs the oracle,
    with shuffled thread order and a sprinkling of invalid frames."""
    from baseband_amd import vdif, synth
    from baseband_amd.vdif.base import VDIFStreamReader
    cfg = dict(cfg)
    nsets = cfg.pop('nsets')
    nthread = cfg['nthread']
    order = list(np.random.default_rng(1).permutation(nthread))
    invalid = [(s, int(s * 7 % nthread)) for s in range(3, nsets, 41)]
    image, h0 = synth.random_vdif(11, nsets, frame_rate=50, thread_order=order,
                                  invalid=invalid, **cfg)
    p = tmp_path / 'synth.vdif'
    p.write_bytes(image.tobytes())
    exp, _ = orc.vdif_read(image, frame_rate=50, fill_value=0.)
    old = VDIFStreamReader.window_bytes
    VDIFStreamReader.window_bytes = 1 << 20        # force many windows
    try:
        with vdif.open(str(p), 'rs', squeeze=False,
                       sample_rate=50 * h0.samples_per_frame) as fh:
            got = fh.read().cpu().numpy()
            assert bits_equal(got, exp)
            fh.seek(h0.samples_per_frame * 5 + 3)
            part = fh.read(h0.samples_per_frame * 40 + 1).cpu().numpy()
            s0 = h0.samples_per_frame * 5 + 3
            assert bits_equal(part, exp[s0:s0 + h0.samples_per_frame * 40 + 1])
    finally:
        VDIFStreamReader.window_bytes = old


def test_mark5b_over_vdif_edv_ab(manifest):
    """EDV 0xab frames carry Mark 5B payloads and use the Mark 5B level
    tables (vdif/payload.py:151-154).  The reference reads them frame by
    frame (its stream reader cannot open such files); here both work."""
    from baseband_amd import vdif
    case = manifest['vdif_edv_ab']
    exp = load_expected('vdif_edv_ab')
    with vdif.open(golden_path(case['file']), 'rb') as fb:
        for i in range(4):
            frame = fb.read_frame()
            assert frame.header.edv == 0xab and frame.nbytes == 10032
            assert frame.shape == (5000, 8)
            assert bits_equal(frame.data.cpu().numpy(), exp[i * 5000:(i + 1) * 5000, 0])
    with vdif.open(golden_path(case['file']), 'rs', sample_rate=32e6) as fh:
        assert fh.shape == (20000, 8)
        assert bits_equal(fh.read().cpu().numpy(), np.ascontiguousarray(exp[:, 0]))


def test_reader_pickle_roundtrip(manifest):
    """Readers are picklable: reopened by name at the saved offset
    (base/base.py:123-151,1020-1032)."""
    import pickle
    from baseband_amd import vdif
    exp = load_expected('sample_vdif')
    with vdif.open(golden_path('samples/sample.vdif'), 'rs', subset=[1, 3]) as fh:
        fh.seek(123)
        blob = pickle.dumps(fh)
        with pickle.loads(blob) as fh2:
            assert fh2.tell() == 123 and fh2.subset == ([1, 3],)
            assert bits_equal(fh2.read(10).cpu().numpy(),
                              np.ascontiguousarray(exp[123:133, [1, 3], 0]))
    with pytest.raises(TypeError):
        import io
        pickle.dumps(vdif.open(io.BytesIO(load_file(manifest['sample_vdif']['file']).tobytes()), 'rs'))


def test_read_into_device_tensor_is_decoded_in_place(manifest):
    """read(out=<device tensor>) on a frame-aligned request decodes straight
    into `out` (no second pass over the output); unaligned requests, subsets
    and host arrays go through a temporary.  Results are the same."""
    import torch
    from baseband_amd import vdif, kernels
    exp = load_expected('sample_vdif')[:, :, 0]
    calls = []
    orig = kernels.VDIFWindow.run

    def spy(self, dbuf, ref, nframes, slots, nsets, within, out, *a, **k):
        calls.append(out)
        return orig(self, dbuf, ref, nframes, slots, nsets, within, out, *a, **k)
    kernels.VDIFWindow.run = spy
    try:
        with vdif.open(golden_path('samples/sample.vdif'), 'rs') as fh:
            out = torch.full((40000, 8), -1., dtype=torch.float32, device='cuda')
            assert fh.read(out=out) is out and fh.tell() == 40000
            assert calls[-1] is not None and calls[-1].data_ptr() == out.data_ptr()
            assert bits_equal(out.cpu().numpy(), exp)
            fh.seek(20000)
            half = torch.empty((20000, 8), dtype=torch.float32, device='cuda')
            fh.read(out=half)
            assert calls[-1].data_ptr() == half.data_ptr()
            assert bits_equal(half.cpu().numpy(), exp[20000:])
            fh.seek(5)                                   # not frame aligned: temporary
            part = torch.empty((100, 8), dtype=torch.float32, device='cuda')
            fh.read(out=part)
            assert calls[-1] is None or calls[-1].data_ptr() != part.data_ptr()
            assert bits_equal(part.cpu().numpy(), exp[5:105])
            fh.seek(0)
            host = np.empty((20000, 8), np.float32)      # host array: copied back
            fh.read(out=host)
            assert bits_equal(host, exp[:20000])
        with vdif.open(golden_path('samples/sample.vdif'), 'rs', subset=[1, 3]) as fh:
            sub = torch.empty((20000, 2), dtype=torch.float32, device='cuda')
            fh.read(out=sub)
            assert bits_equal(sub.cpu().numpy(), exp[:20000][:, [1, 3]])
    finally:
        kernels.VDIFWindow.run = orig


def test_vdif_frame_from_mark5b_frame_matches_reference():
    """VDIFFrame.from_mark5b_frame: EDV 0xab frames around the sample Mark 5B
    frames, byte-identical to what the reference builds (vdif/frame.py:104-128)."""
    import io
    import json
    import hashlib
    from baseband_amd import mark5b
    from baseband_amd.vdif import VDIFFrame
    with open(golden_path('header_fuzz_cases.json')) as f:
        gold = json.load(f)['mark5b_to_vdif']
    with mark5b.open(golden_path('samples/sample.m5b'), 'rb', kday=56000, nchan=8) as fh:
        for want in gold:
            m5 = fh.read_frame()
            vf = VDIFFrame.from_mark5b_frame(m5)
            b = io.BytesIO()
            vf.tofile(b)
            assert hashlib.sha256(b.getvalue()).hexdigest() == want['frame_sha256']
            assert vf.valid == want['valid']
            assert bool((vf.data == m5.data).all())


def test_frames_and_frame_sets_convert_to_numpy(manifest):
    """``numpy.asarray(frame)`` / ``asarray(frameset)`` bring the decoded data
    to the host like the reference's ``FrameBase.__array__``
    (base/frame.py:182-187)."""
    from baseband_amd import vdif
    with vdif.open(golden_path('samples/sample.vdif'), 'rb') as fh:
        frame = fh.read_frame()
        fh.seek(0)
        frameset = fh.read_frameset()
    a = np.asarray(frame)
    assert a.dtype == np.float32 and a.shape == tuple(frame.shape)
    assert bits_equal(a, frame.data.cpu().numpy())
    s = np.asarray(frameset)
    assert s.shape == tuple(frameset.data.shape) and bits_equal(s, frameset.data.cpu().numpy())
    assert np.asarray(frame, dtype=np.float64).dtype == np.float64


def test_read_into_numpy_and_asnumpy(tmp_path):
    """``read(out=<ndarray>)`` for a large request and ``baseband_amd.asnumpy``
    go through the pinned double-buffered download; the samples must equal the
    device result."""
    import torch
    import baseband_amd
    from baseband_amd import vdif, synth
    image, h0 = synth.random_vdif(11, 400, payload_nbytes=8000, frame_rate=1000)
    path = str(tmp_path / 'x.vdif')
    image.tofile(path)
    with vdif.open(path, 'rs', sample_rate=32e6) as fh:
        dev = fh.read()
        fh.seek(0)
        host = np.empty(tuple(dev.shape), np.float32)
        assert host.nbytes > (32 << 20)
        got = fh.read(out=host)
        assert got is host and bits_equal(host, dev.cpu().numpy())
        fh.seek(5)
        small = np.empty((1000,), np.float32)
        fh.read(out=small)
        assert bits_equal(small, dev[5:1005].cpu().numpy())
    assert bits_equal(baseband_amd.asnumpy(dev), dev.cpu().numpy())
    z = torch.view_as_complex(torch.randn(100, 3, 2, device='cuda'))
    assert np.array_equal(baseband_amd.asnumpy(z), z.cpu().numpy())
    strided = np.empty((dev.shape[0], 2), np.float32)[:, 0]
    assert bits_equal(baseband_amd.asnumpy(dev, out=strided), dev.cpu().numpy())


@pytest.mark.parametrize('squeeze,subset,folded', [
    (True, (slice(None), [3, 1, 14]), True), (True, (slice(None), slice(2, 12, 3)), True),
    (True, ([6, 1], slice(3, 9)), True), (True, (2, 5), True), (False, (slice(None), 7), True),
    (True, ([5], slice(None)), False),           # threads only: nothing to fold
    (True, (slice(None), [15]), True), (False, ([2], [3]), True)])
def test_channel_subset_is_folded_into_the_decode(manifest, squeeze, subset, folded):
    """A `subset` that picks channels is applied BY the decode kernel
    (bb_decode_frames_select): same values as NumPy indexing of the full
    reference output (base/base.py:706-717), whole-stream and partial reads,
    read(out=...) straight into the caller's tensor."""
    import torch
    from baseband_amd import vdif, _lib
    case = manifest['vdif_cfg3_small']
    exp = load_expected('vdif_cfg3_small')       # (4000, 8, 16) c64
    full = exp
    want = full[(slice(None),) + subset]        # (8 threads, 16 channels: nothing to squeeze first)
    with vdif.open(golden_path(case['file']), 'rs', squeeze=squeeze, subset=subset, **_kw(case)) as fh:
        assert (fh._within_np is not None) == folded
        assert fh.sample_shape == want.shape[1:]
        got = fh.read()
        if folded:
            # (k_decode_pick for selections of up to an eighth of a thread sample, k_decode_gather_select beyond)
            assert 'k_decode_gather_select' in _lib.last_kernel() or 'k_decode_pick' in _lib.last_kernel()
        assert bits_equal(got.cpu().numpy(), np.ascontiguousarray(want))
        fh.seek(995)
        assert bits_equal(fh.read(1010).cpu().numpy(), np.ascontiguousarray(want[995:2005]))
        out = torch.empty((2000,) + want.shape[1:], dtype=torch.complex64, device='cuda')
        fh.seek(1000)
        assert fh.read(out=out) is out
        assert bits_equal(out.cpu().numpy(), np.ascontiguousarray(want[1000:3000]))
