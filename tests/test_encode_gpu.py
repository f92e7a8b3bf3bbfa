"""GPU encoders (float32 -> packed codes) against the reference encoders'
outputs (tests/golden/encode_cases.npz) and the oracle."""
import json

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import golden_path, bits_equal

pytestmark = pytest.mark.gpu

CODERS = {'vdif': 0, 'mark5b': 1, 'int': 2}
CASES = [('vdif', 1, 'vdif1'), ('vdif', 2, 'vdif2'), ('vdif', 4, 'vdif4'), ('vdif', 8, 'vdif8'),
         ('mark5b', 1, 'mark5b1'), ('mark5b', 2, 'mark5b2'), ('int', 4, 'int4'), ('int', 8, 'int8')]


@pytest.fixture(scope='module')
def gold():
    return np.load(golden_path('encode_cases.npz'))


@pytest.mark.parametrize('coder,bps,key', CASES)
def test_flat_encoders_match_reference(gold, coder, bps, key):
    import torch
    from baseband_amd import kernels
    x = torch.from_numpy(gold['input']).cuda()
    got = kernels.encode_flat(x, CODERS[coder], bps).cpu().numpy()
    assert np.array_equal(got, gold[key])


@pytest.mark.parametrize('coder,bps,key', CASES)
def test_flat_encoders_large_random_vs_oracle(coder, bps, key):
    import torch
    from baseband_amd import kernels
    rng = np.random.default_rng(bps * 7 + len(coder))
    scale = {1: 1.0, 2: 2.2, 4: 1.4, 8: 1.3}[bps] * (30. if coder == 'int' and bps == 8 else 1.)
    x = (rng.standard_normal(1 << 20) * scale).astype(np.float32)
    got = kernels.encode_flat(torch.from_numpy(x).cuda(), CODERS[coder], bps).cpu().numpy()
    want = orc.encode_flat(x, coder, bps) if bps < 8 else orc.encode_codes(x, coder, bps)
    assert np.array_equal(got, want)
    # decode(encode(x)) == nearest level, and re-encoding is idempotent
    dec = kernels.decode_frames(torch.from_numpy(want).cuda(), 1, want.size, CODERS[coder], bps)
    again = kernels.encode_flat(dec, CODERS[coder], bps).cpu().numpy()
    assert np.array_equal(again, want)


def test_two_bit_threshold_encoder_equals_direct_arithmetic_for_every_float():
    """Default 2-bit path (three compares against host-bisected thresholds) vs
    the per-sample clip/add/floor_divide arithmetic, over all 2^32 float32 bit
    patterns except NaNs (whose integer cast numpy leaves undefined)."""
    import torch
    from baseband_amd import kernels, _lib
    chunk = 1 << 28
    try:
        for start in range(0, 1 << 32, chunk):
            bits = torch.arange(start, start + chunk, dtype=torch.int64, device='cuda')
            x = (bits - ((bits >> 31) << 32)).to(torch.int32).view(torch.float32)
            x = torch.where(torch.isnan(x), torch.zeros_like(x), x)
            del bits
            kernels.tune(_lib.TUNE_ENCODE_DIRECT, 0)
            fast = kernels.encode_flat(x, 0, 2)
            fast5 = kernels.encode_flat(x, 1, 2)
            kernels.tune(_lib.TUNE_ENCODE_DIRECT, 1)
            direct = kernels.encode_flat(x, 0, 2)
            direct5 = kernels.encode_flat(x, 1, 2)
            assert torch.equal(fast, direct) and torch.equal(fast5, direct5)
            del x, fast, direct, fast5, direct5
    finally:
        kernels.tune(_lib.TUNE_ENCODE_DIRECT, 0)


@pytest.mark.parametrize('direct', [0, 1])
def test_two_bit_encoder_near_steps_vs_oracle(direct):
    import torch
    from baseband_amd import kernels, _lib
    thr = _lib.encode_thresholds()
    xs = []
    for t in list(thr) + [np.float32(-3.261846), np.float32(3.261846), np.float32(0)]:
        u = np.arange(-4096, 4096, dtype=np.int64) + int(np.float32(t).view(np.int32))
        xs.append(u.astype(np.int32).view(np.float32))
    xs.append(np.array([np.inf, -np.inf, 1e38, -1e38, 1e-45, -1e-45, -0.0, 0.0], np.float32))
    x = np.concatenate(xs)
    x = x[:x.size // 256 * 256 + 4 * 7]       # whole runs plus a ragged tail of 7 quads
    try:
        kernels.tune(_lib.TUNE_ENCODE_DIRECT, direct)
        for coder in ('vdif', 'mark5b'):
            got = kernels.encode_flat(torch.from_numpy(x).cuda(), CODERS[coder], 2).cpu().numpy()
            assert np.array_equal(got, orc.encode_flat(x, coder, 2))
    finally:
        kernels.tune(_lib.TUNE_ENCODE_DIRECT, 0)


@pytest.mark.parametrize('coder,bps,key', CASES)
@pytest.mark.parametrize('n', [4 * 8, 4 * 255, 4 * 256, 4 * 257 + 8, 4 * 1024 * 3 + 16])
def test_flat_encoders_run_and_tail_sizes(coder, bps, key, n):
    import torch
    from baseband_amd import kernels
    n = n // (32 // bps) * (32 // bps) if bps < 8 else n
    x = (np.random.default_rng(n + bps).standard_normal(n) * 2.5).astype(np.float32)
    got = kernels.encode_flat(torch.from_numpy(x).cuda(), CODERS[coder], bps).cpu().numpy()
    want = orc.encode_flat(x, coder, bps) if bps < 8 else orc.encode_codes(x, coder, bps)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_complex_and_errors():
    import torch
    from baseband_amd import kernels, _lib
    z = torch.tensor([1 - 3j, 0.5 + 9j, -0.1 - 0.2j, 2.2 + 2.1j], dtype=torch.complex64, device='cuda')
    got = kernels.encode_flat(z, 0, 2).cpu().numpy()
    want = orc.encode_flat(np.array([1, -3, .5, 9, -.1, -.2, 2.2, 2.1], np.float32), 'vdif', 2)
    assert np.array_equal(got, want)
    with pytest.raises(KeyError):
        kernels.encode_flat(torch.zeros(8, device='cuda'), _lib.CODER_MARK5B, 4)
    with pytest.raises(_lib.BBError):
        kernels.encode_flat(torch.zeros(6, device='cuda'), 0, 2)       # not whole bytes
    # whole bytes that are not whole quads are padded on the host side
    got = kernels.encode_flat(torch.tensor([1., -3.], device='cuda'), _lib.CODER_INT, 4).cpu().numpy()
    assert got.tolist() == [0xd1]


def test_mark4_encoders_match_reference(gold):
    import torch
    from baseband_amd import kernels
    with open(golden_path('mark4_bitmaps.json')) as f:
        maps = json.load(f)
    x = gold['input']
    for name, e in maps.items():
        nchan, fanout, nt = e['nchan'], e['fanout'], e['ntrack']
        n = nchan * fanout * (8192 // (nchan * fanout))
        data = torch.from_numpy(x[:n].reshape(-1, nchan).copy()).cuda()
        got = kernels.encode_mark4(data, nt, e['sign_bit'], e['mag_bit']).cpu().numpy()
        assert np.array_equal(got, gold['mark4_' + name]), name
        # and back
        dbuf = torch.from_numpy(np.concatenate([got, np.zeros(8, np.uint8)])).cuda()
        dec = kernels.decode_mark4(dbuf, 1, nt, got.size // (nt // 8), e['sign_bit'], e['mag_bit'])
        lev = orc.LEVELS_2[orc.encode_codes(x[:n], 'vdif', 2)]
        assert bits_equal(dec.cpu().numpy(), lev)


VDIF_WRITE_CASES = ['vdif_cfg2_small', 'vdif_cfg3_small', 'vdif_bps1_c4', 'vdif_bps4_cplx_t2',
                    'vdif_bps8_real_c2', 'vdif_bps8_cplx_t4', 'vdif_bps2_t8_c1',
                    'vdif_legacy_bps2', 'vdif_bps4_t2_c1']


@pytest.mark.parametrize('name', VDIF_WRITE_CASES)
def test_vdif_stream_writer_is_byte_identical_to_reference(manifest, name):
    """open(..., 'ws').write(data) produces the file the reference's stream
    writer produced for the same data (tests/golden/synth/*.bin)."""
    import io
    import torch
    from conftest import load_expected, load_file
    from baseband_amd import vdif
    case = manifest[name]
    data = load_expected(name)
    blob = load_file(case['file'])
    kw = dict(edv=case['edv'], bps=case['bps'], nchan=case['nchan'],
              complex_data=case['complex_data'], station='AA',
              time=np.datetime64('2020-01-01T00:00:00'))
    if case['edv'] == 3:
        kw['frame_length'] = 629
    else:
        kw['samples_per_frame'] = case['samples_per_frame']
    out = io.BytesIO()
    out.close = lambda: None                     # keep the buffer readable
    fw = vdif.open(out, 'ws', sample_rate=case['frame_rate'] * case['samples_per_frame'],
                   nthread=case['nthread'], squeeze=False, **kw)
    # feed it in uneven pieces, from the host and from the device
    n = data.shape[0]
    cuts = [0, 7, n // 3 + 1, n // 2, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        piece = data[a:b]
        fw.write(torch.from_numpy(piece).cuda() if a % 2 else piece)
    assert fw.tell() == n
    fw.close()
    mine = np.frombuffer(out.getvalue(), np.uint8)
    nth = case['nthread']
    if nth > 1:                                  # the fixture stores threads as 1,3,..,0,2,..
        fn = len(blob) // (case['nframes'] * nth)
        perm = list(range(1, nth, 2)) + list(range(0, nth, 2))
        mine = mine.reshape(case['nframes'], nth, fn)[:, perm].reshape(-1)
    assert mine.tobytes() == blob.tobytes()


def test_vdif_writer_partial_last_frame_and_roundtrip(manifest, tmp_path):
    from baseband_amd import vdif
    rng = np.random.default_rng(5)
    lev = orc.LEVELS_2
    data = lev[rng.integers(0, 4, size=(2500, 2, 4))].astype(np.float32)
    p = str(tmp_path / 'w.vdif')
    with pytest.warns(UserWarning, match='partial buffer'):
        with vdif.open(p, 'ws', sample_rate=100000., nthread=2, edv=0, bps=2, nchan=4,
                       samples_per_frame=1000, station='ab',
                       time=np.datetime64('2021-03-04T05:06:07')) as fw:
            fw.write(data[:1234])
            fw.write(data[1234:], valid=True)
    with vdif.open(p, 'rs', sample_rate=100000., squeeze=False, verify=False) as fr:
        assert fr.shape == (3000, 2, 4)
        assert str(fr.start_time).startswith('2021-03-04T05:06:07')
        back = fr.read().cpu().numpy()
    assert bits_equal(back[:2000], data[:2000])
    assert np.all(back[2000:] == 0.)             # padded frame is flagged invalid -> fill


@pytest.mark.parametrize('name', ['m5b_c16_b2', 'm5b_c8_b1', 'm5b_c4_b2'])
def test_mark5b_stream_writer_is_byte_identical_to_reference(manifest, name):
    import io
    from conftest import load_expected, load_file
    from baseband_amd import mark5b
    case = manifest[name]
    data = load_expected(name)
    blob = load_file(case['file'])
    out = io.BytesIO()
    out.close = lambda: None
    with mark5b.open(out, 'ws', sample_rate=case['frame_rate'] * case['samples_per_frame'],
                     nchan=case['nchan'], bps=case['bps'],
                     time=np.datetime64('2014-06-13T05:30:01')) as fw:
        fw.write(data)
    mine = np.frombuffer(out.getvalue(), np.uint8).reshape(-1, 10016)
    ref = blob.reshape(-1, 10016)
    for f in range(case['nframes']):
        if f in case['invalid']:
            assert mine[f, :16].tobytes() == ref[f, :16].tobytes()   # payload was replaced in the fixture
        else:
            assert mine[f].tobytes() == ref[f].tobytes(), f


@pytest.mark.parametrize('name', ['m4_t64_f4', 'm4_t32_f4', 'm4_t32_f2', 'm4_t16_f4'])
def test_mark4_stream_writer_is_byte_identical_to_reference(manifest, name):
    import io
    from conftest import load_expected, load_file
    from baseband_amd import mark4
    case = manifest[name]
    data = load_expected(name)
    blob = load_file(case['file'])
    out = io.BytesIO()
    out.close = lambda: None
    with mark4.open(out, 'ws', sample_rate=case['frame_rate'] * case['samples_per_frame'],
                    ntrack=case['ntrack'], fanout=case['fanout'], bps=2,
                    time=np.datetime64(case['start_time'])) as fw:
        fw.write(data[:1000])
        fw.write(data[1000:])
    fn = case['ntrack'] * 2500
    mine = np.frombuffer(out.getvalue(), np.uint8)
    assert len(mine) == len(blob)
    for f in range(case['nframes']):
        if f in case['invalid']:
            continue            # the fixture's error-flag frame decoded to fill
        assert mine[f * fn:(f + 1) * fn].tobytes() == blob[f * fn:(f + 1) * fn].tobytes(), f


def test_payload_fromdata_on_device_roundtrip(manifest):
    """Payload.fromdata packs on the GPU whether the samples come as device
    tensors or as host arrays (uploaded first); the words equal what the
    input-synthesis encoders of synth_codes produce and decode back to the data."""
    import torch
    from conftest import load_expected
    from baseband_amd.vdif import VDIFPayload, VDIFHeader
    from baseband_amd.mark4 import Mark4Payload, Mark4Header
    data = load_expected('vdif_cfg3_small')[:1000, 0]              # (1000, 16) complex
    h = VDIFHeader(manifest['vdif_cfg3_small']['header0_words'])
    dev = VDIFPayload.fromdata(torch.from_numpy(data).cuda(), header=h)
    host = VDIFPayload.fromdata(data, header=h)
    assert dev == host and np.array_equal(dev.words, host.words)
    from baseband_amd import synth_codes
    want = synth_codes.pack_codes(synth_codes.codes_2bit(synth_codes.components(data)), 2)
    assert np.array_equal(dev.words.view(np.uint8), want)
    assert bits_equal(dev.data.cpu().numpy(), data)
    pl = VDIFPayload.fromdata(torch.from_numpy(data.real.copy()).cuda(), bps=4)
    assert pl.shape == (1000, 16) and pl.bps == 4
    case = manifest['m4_t32_f2']
    m4 = load_expected('m4_t32_f2')
    h4 = Mark4Header(np.array(case['header0_words'], np.uint32), decade=2010)
    body = m4[160 * 2:40000]
    pd = Mark4Payload.fromdata(torch.from_numpy(body).cuda(), h4)
    ph = Mark4Payload.fromdata(body, h4)
    assert np.array_equal(pd.words, ph.words)
    assert np.array_equal(pd.words, synth_codes.encode_mark4(body, h4))
    assert bits_equal(pd.data.cpu().numpy(), body)
    # the block formats too: GUPPI storage orders, DADA / MKBF
    from baseband_amd.guppi import GUPPIPayload
    from baseband_amd.dada import DADAPayload
    rng = np.random.default_rng(8)
    z = (rng.integers(-128, 128, (64, 2, 4)) + 1j * rng.integers(-128, 128, (64, 2, 4))).astype(np.complex64)
    for cf in (True, False):
        gp = GUPPIPayload.fromdata(z, bps=8, channels_first=cf)
        assert gp.words.dtype == np.int8 and bits_equal(gp.data.cpu().numpy(), z)
    dp = DADAPayload.fromdata(z[:, :, :1] + np.complex64(0.3 - 0.3j), bps=8)   # rounds back
    assert bits_equal(dp.data.cpu().numpy(), z[:, :, :1])
    with pytest.raises(ValueError):
        VDIFPayload.fromdata(torch.zeros(8, 1, device='cuda'), bps=3)


# ---- block formats: DADA / GUPPI stream writers -----------------------------
@pytest.fixture(scope='module')
def block_gold():
    return np.load(golden_path('block_writer_cases.npz'))


@pytest.mark.parametrize('key', ['dada', 'dada_real', 'guppi_cf', 'guppi_tf'])
def test_block_stream_writers_are_byte_identical_to_reference(block_gold, key, tmp_path):
    """Non-integer, out-of-range samples in pieces, partial last frame padded
    at close: the file must equal what the reference's writer produced
    (oracle/gen_golden.py `block_writers`)."""
    import io
    import warnings
    import importlib
    fmt = key.split('_')[0]
    mod = importlib.import_module('baseband_amd.' + fmt)
    want = block_gold[key + '_file'].tobytes()
    data = block_gold[key + '_in']
    header_class = mod.DADAHeader if fmt == 'dada' else mod.GUPPIHeader
    header0 = header_class.fromfile(io.BytesIO(want))
    name = str(tmp_path / ('w.' + fmt))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with mod.open(name, 'ws', header0=header0) as fw:
            assert fw.sample_shape == tuple(data.shape[1:])
            fw.write(data[:50])
            fw.write(data[50:])
            assert fw.tell() == len(data)
    got = open(name, 'rb').read()
    assert len(got) == len(want)
    assert got == want
    with mod.open(name, 'rs') as fr:
        back = fr.read().cpu().numpy()
    if key == 'dada':
        assert bits_equal(back, block_gold['dada_back'])
    # int8 round/clip semantics: reading back gives clip(rint(x))
    comp = np.ascontiguousarray(data).view(np.float32)
    expect = np.clip(np.rint(comp), -128, 127).astype(np.float32)
    assert np.array_equal(np.ascontiguousarray(back[:len(data)]).view(np.float32).reshape(expect.shape), expect)
    assert not back[len(data):].any()


def test_dada_sequence_writer_matches_reference(tmp_path):
    """One frame per file through the {obs_offset} template; names, sizes and
    bytes as the reference writes them (tests/golden/sequence_cases.json)."""
    from baseband_amd import dada
    with open(golden_path('sequence_cases.json')) as f:
        g = json.load(f)['dada_sequence']
    import hashlib
    import torch
    with dada.open(golden_path('samples/sample.dada'), 'rs') as f1:
        data = f1.read()
        header0 = f1.header0
    template = str(tmp_path / '{utc_start}.{obs_offset:016d}.000000.dada')
    with dada.open(template, 'ws', header0=header0) as fw:
        fw.write(data)
        fw.write(data)
    import os
    files = sorted(os.listdir(str(tmp_path)))
    assert files == g['files']
    assert [os.path.getsize(str(tmp_path / f)) for f in files] == g['sizes']
    assert [hashlib.sha256(open(str(tmp_path / f), 'rb').read()).hexdigest() for f in files] == g['sha256']
    with pytest.raises(KeyError):
        dada.open(template, 'rs')                          # UTC_START unknown
    with dada.open(template, 'rs', UTC_START=header0['UTC_START'],
                   OBS_OFFSET=header0['OBS_OFFSET'], FILE_SIZE=header0['FILE_SIZE']) as fr:
        assert fr.shape == tuple(g['shape'])
        back = fr.read()
    assert hashlib.sha256(back.cpu().numpy().tobytes()).hexdigest() == g['data_sha256']
    assert bool((back == torch.cat([data, data])).all())
    # a list of the same names works too, and a single member is a stream of its own
    with dada.open([str(tmp_path / f) for f in files], 'rs') as fr:
        assert bool((fr.read() == torch.cat([data, data])).all())
    with dada.open(str(tmp_path / files[1]), 'rs') as fr:
        assert fr.start_time > header0.time and bool((fr.read() == data).all())


def test_guppi_sequence_writer_frames_per_file(block_gold, tmp_path):
    import io
    from baseband_amd import guppi
    want = block_gold['guppi_cf_file'].tobytes()
    data = block_gold['guppi_cf_in']
    header0 = guppi.GUPPIHeader.fromfile(io.BytesIO(want))
    names = [str(tmp_path / ('g%d.raw' % i)) for i in range(2)]
    with guppi.open(names, 'ws', header0=header0, frames_per_file=2) as fw:
        fw.write(data)
    fn = header0.frame_nbytes
    assert open(names[0], 'rb').read() == want[:2 * fn]
    assert open(names[1], 'rb').read() == want[2 * fn:]
    with guppi.open(names, 'rs') as fr, guppi.open(io.BytesIO(want), 'rs') as f1:
        assert bits_equal(fr.read().cpu().numpy(), f1.read().cpu().numpy())


# ---- frame-level ('wb') writers ---------------------------------------------
@pytest.mark.parametrize('fmt,name,kw', [
    ('vdif', 'vdif_cfg3_small', {}), ('vdif', 'vdif_bps8_real_c2', {}),
    ('vdif', 'vdif_bps4_cplx_t2', {}), ('vdif', 'vdif_bps1_c4', {}), ('vdif', 'vdif_edv_ab', {}),
    ('mark5b', 'm5b_c16_b2', dict(nchan=16, kday=58000)),
    ('mark5b', 'm5b_c8_b1', dict(nchan=8, kday=58000, bps=1)),
    ('mark4', 'm4_t64_f4', dict(ntrack=64, decade=2010)),
    ('mark4', 'm4_t32_f2', dict(ntrack=32, decade=2010)),
    ('mark4', 'm4_t16_f4', dict(ntrack=16, decade=2010)),
    ('guppi', 'guppi_cf_c64_ov32', {}), ('guppi', 'guppi_tf_c8_ov16', {}),
    ('guppi', 'guppi_real_c1', {}), ('guppi', 'guppi_cf_c6_p1', {}),
    ('dada', 'dada_p1_c1_real', {})])
def test_file_writers_reproduce_reference_files(manifest, fmt, name, kw, tmp_path):
    """Every frame of a file written by the reference is read ('rb'), decoded on
    the GPU, and written again through the 'wb' writer from samples + header:
    the new file equals the reference's file byte for byte."""
    import importlib
    mod = importlib.import_module('baseband_amd.' + fmt)
    src = golden_path(manifest[name]['file'])
    want = open(src, 'rb').read()
    out = str(tmp_path / 'copy.bin')
    with mod.open(src, 'rb', **kw) as fr, mod.open(out, 'wb') as fw:
        if fmt in ('mark5b', 'mark4'):
            fr.find_header()
        lead = fr.tell()
        while fr.tell() < len(want):
            frame = fr.read_frame()
            if fmt == 'mark5b':
                fw.write_frame(frame.data, frame.header, bps=frame.payload.bps, valid=frame.valid)
            elif not frame.valid:
                fw.write_frame(frame)          # samples of an invalid frame are not recoverable
            else:
                fw.write_frame(frame.data, frame.header)
    assert open(out, 'rb').read() == want[lead:]
    with pytest.raises(ValueError):
        mod.open([out, out + '2'], 'wb')                   # no sequences in binary mode


def test_vdif_write_frameset(manifest, tmp_path):
    """write_frameset from (samples, thread, chan) data and one header: frames
    for threads 0..n-1 in order (vdif/frame.py:250-311)."""
    from baseband_amd import vdif
    src = golden_path(manifest['vdif_bps8_cplx_t4']['file'])
    out = str(tmp_path / 'sets.vdif')
    with vdif.open(src, 'rb') as fr, vdif.open(out, 'wb') as fw:
        sets = []
        while True:
            try:
                sets.append(fr.read_frameset())
            except EOFError:
                break
        for fs in sets:
            h = fs.frames[0].header.copy()
            h['thread_id'] = 0
            fw.write_frameset(fs.data, h)
    with vdif.open(src, 'rs', squeeze=False, sample_rate=25600.) as f1, \
            vdif.open(out, 'rs', squeeze=False, sample_rate=25600.) as f2:
        assert f1.shape == f2.shape == (1536, 4, 1) and bool((f1.read() == f2.read()).all())
    with vdif.open(out, 'rb') as fr:
        assert [fr.read_frame().header['thread_id'] for _ in range(8)] == [0, 1, 2, 3] * 2


# ---- GSB stream writer --------------------------------------------------------
@pytest.fixture(scope='module')
def gsb_gold():
    return np.load(golden_path('gsb_writer_cases.npz'))


def test_gsb_writer_rewrites_the_sample_observations(gsb_gold, tmp_path):
    """Reading the two sample observations and writing them again gives the
    timestamp text and raw bytes the reference's writer gives
    (oracle/gen_golden.py `gsb_writer`)."""
    from baseband_amd import gsb
    d = golden_path('samples/gsb/')
    ts, raw = str(tmp_path / 'r.timestamp'), str(tmp_path / 'r.dat')
    with gsb.open(d + 'sample_gsb_rawdump.timestamp', 'rs', raw=d + 'sample_gsb_rawdump.dat',
                  samples_per_frame=8192) as fr:
        data = fr.read()
        with gsb.open(ts, 'ws', raw=raw, header0=fr.header0, sample_rate=fr.sample_rate,
                      samples_per_frame=8192) as fw:
            fw.write(data[:10000])
            fw.write(data[10000:])
            assert fw.tell() == data.shape[0]
    assert open(ts, 'rb').read() == gsb_gold['rawdump_ts'].tobytes()
    assert open(raw, 'rb').read() == gsb_gold['rawdump_raw'].tobytes()
    names = [[d + 'sample_gsb_phased.Pol-%s%d.dat' % (pol, part) for part in (1, 2)] for pol in 'LR']
    import os
    if not os.path.exists(names[0][0]):
        names = [[d + f for f in sorted(os.listdir(d)) if 'phased' in f and f.endswith('.dat')][i:i + 2]
                 for i in (0, 2)]
    ts = str(tmp_path / 'p.timestamp')
    raws = [[str(tmp_path / ('p%d%d.dat' % (p, f))) for f in range(2)] for p in range(2)]
    with gsb.open(d + 'sample_gsb_phased.timestamp', 'rs', raw=names, samples_per_frame=8) as fr:
        data = fr.read()
        with gsb.open(ts, 'ws', raw=raws, header0=fr.header0, sample_rate=fr.sample_rate,
                      samples_per_frame=8) as fw:
            fw.write(data)
    assert open(ts, 'rb').read() == gsb_gold['phased_ts'].tobytes()
    for p in range(2):
        for f in range(2):
            assert open(raws[p][f], 'rb').read() == gsb_gold['phased_raw%d%d' % (p, f)].tobytes()
    with gsb.open(ts, 'rs', raw=raws, samples_per_frame=8) as fb:
        assert bool((fb.read() == data).all())


def test_gsb_writer_from_keywords(gsb_gold, tmp_path):
    """Headers from keywords, non-integer samples (4-bit clipping, int8
    rounding), sequence numbers growing a digit, memory block modulo 8."""
    from baseband_amd import gsb
    t0 = np.datetime64('2015-06-01T01:02:03.251658240')
    ts, raw = str(tmp_path / 'r2.timestamp'), str(tmp_path / 'r2.dat')
    with gsb.open(ts, 'ws', raw=raw, time=t0, samples_per_frame=64, sample_rate=1e3) as fw:
        assert fw.header0.mode == 'rawdump' and fw.bps == 4 and fw.sample_shape == ()
        fw.write(gsb_gold['raw2_in'])
    assert open(ts, 'rb').read() == gsb_gold['raw2_ts'].tobytes()
    assert open(raw, 'rb').read() == gsb_gold['raw2_raw'].tobytes()
    ts, raw = str(tmp_path / 'p2.timestamp'), str(tmp_path / 'p2.dat')
    with gsb.open(ts, 'ws', raw=raw, header_mode='phased', time=t0, seq_nr=9998, mem_block=6,
                  samples_per_frame=8, nchan=4, sample_rate=2e3) as fw:
        assert fw.header0.mode == 'phased' and fw.bps == 8 and fw.complex_data
        fw.write(gsb_gold['ph2_in'])
    assert open(ts, 'rb').read() == gsb_gold['ph2_ts'].tobytes()
    assert open(raw, 'rb').read() == gsb_gold['ph2_raw'].tobytes()
    with pytest.raises(TypeError):
        gsb.open(ts, 'ws', time=t0)                              # no raw
    with pytest.raises(TypeError):
        gsb.open(ts, 'wb', raw=raw)                              # raw is for streams


def test_base_encoding_module_matches_oracle_and_library_tables(gold):
    """``baseband_amd.base.encoding`` (the names of base/encoding.py:12-160):
    level tables are the library's and equal the reference's; the unpacked
    coders and the 8-bit decoder equal the oracle for odd-sized inputs."""
    import torch
    from baseband_amd.base import encoding as enc
    for bps in (1, 2, 4):
        assert bits_equal(enc.decoder_levels[bps], orc.code_levels('vdif', bps))
    with pytest.raises(KeyError):
        enc.decoder_levels[3]
    assert (enc.OPTIMAL_2BIT_HIGH, enc.TWO_BIT_1_SIGMA, enc.FOUR_BIT_1_SIGMA,
            enc.EIGHT_BIT_1_SIGMA) == (3.316505, 2.174564, 2.95, 35.5)
    rng = np.random.default_rng(77)
    x = (rng.standard_normal((37, 3)) * 2.5).astype(np.float32)       # 111 samples: ragged
    for bps, fn in ((1, enc.encode_1bit_base), (2, enc.encode_2bit_base),
                    (4, enc.encode_4bit_base), (8, enc.encode_8bit)):
        scale = 30. if bps == 8 else 1.
        got = fn(x[:36] * np.float32(scale))
        want = orc.encode_codes((x[:36] * np.float32(scale)).ravel(), 'vdif', bps).reshape(36, 3)
        assert got.shape == (36, 3) and got.dtype == torch.uint8
        assert np.array_equal(got.cpu().numpy(), want), bps
    words = rng.integers(0, 256, 1003, dtype=np.uint8)
    out = enc.decode_8bit(words)
    assert bits_equal(out.cpu().numpy(), orc.code_levels('vdif', 8)[words])
    out2 = enc.decode_8bit(torch.from_numpy(words).cuda())
    assert bool((out2 == out).all())


def test_large_host_arrays_are_written_like_device_tensors(tmp_path):
    """A large NumPy input goes to the GPU through the pinned upload
    (`staging.upload_array`); the file must equal the one written from the
    same samples already on the device."""
    import torch
    from baseband_amd import vdif
    from baseband_amd.vdif.header import VDIFHeader
    from baseband_amd.staging import upload_array
    rng = np.random.default_rng(21)
    data = (rng.standard_normal(32000 * 300) * 2.).astype(np.float32)      # 38 MB
    assert data.nbytes > (32 << 20)
    dev = upload_array(data)
    assert dev.is_cuda and dev.dtype == torch.float32 and bool((dev.cpu() == torch.from_numpy(data)).all())
    z = (rng.standard_normal((5_000_000, 2)) + 0j).astype(np.complex64)
    assert bool((upload_array(z).cpu() == torch.from_numpy(z)).all())
    h0 = VDIFHeader.fromvalues(edv=0, time=np.datetime64('2014-06-13T05:30:01'), nchan=1, bps=2,
                               complex_data=False, thread_id=0, samples_per_frame=32000, station='AA')
    a, b = str(tmp_path / 'a.vdif'), str(tmp_path / 'b.vdif')
    with vdif.open(a, 'ws', header0=h0, sample_rate=32e6, nthread=1) as fw:
        fw.write(data)
    with vdif.open(b, 'ws', header0=h0, sample_rate=32e6, nthread=1) as fw:
        fw.write(torch.from_numpy(data).cuda())
    assert open(a, 'rb').read() == open(b, 'rb').read()
