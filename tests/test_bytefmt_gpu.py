"""GUPPI, DADA (incl. MKBF) and GSB through the drop-in API on the GPU,
bit-exact vs the reference's outputs."""
import hashlib

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import golden_path, load_expected, load_file, bits_equal

pytestmark = pytest.mark.gpu

GUPPI_CASES = ['sample_puppi', 'guppi_cf_c64_ov0', 'guppi_cf_c64_ov32',
               'guppi_tf_c8_ov16', 'guppi_cf_c6_p1', 'guppi_real_c1']
DADA_CASES = ['sample_dada', 'sample_meerkat_dada', 'sample_mkbf_dada',
              'dada_p2_c4_cplx', 'dada_p1_c1_real', 'dada_p2_c3_real']


def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


@pytest.mark.parametrize('name', GUPPI_CASES)
def test_guppi_stream(manifest, name):
    from baseband_amd import guppi
    case = manifest[name]
    exp = load_expected(name)
    with guppi.open(golden_path(case['file']), 'rs', squeeze=False) as fh:
        assert bits_equal(fh.read().cpu().numpy(), exp)
        for off, cnt, digest in case['reads']:      # overlap rules of the reference loop
            fh.seek(off)
            assert _sha(fh.read(cnt)) == digest, (off, cnt)
        fh.seek(0)
        with pytest.raises(EOFError):
            fh.read(exp.shape[0] + 1)
    with guppi.open(golden_path(case['file']), 'rs') as fh:
        got = fh.read(7).cpu().numpy()
        want = exp[:7].reshape((7,) + tuple(s for s in exp.shape[1:] if s > 1))
        assert bits_equal(got, np.ascontiguousarray(want))


def test_guppi_known_answers_and_subset(manifest):
    """guppi/tests/test_guppi.py:236-249,504-510."""
    from baseband_amd import guppi
    exp = load_expected('sample_puppi')
    with guppi.open(golden_path('samples/sample_puppi.raw'), 'rs', subset=(0, [1, 3])) as fh:
        assert fh.sample_shape == (2,)
        got = fh.read(3).cpu().numpy()
    assert bits_equal(got, np.ascontiguousarray(exp[:3, 0][:, [1, 3]]))
    assert exp[0, 0, 0] == -7 + 12j


@pytest.mark.parametrize('name', ['sample_puppi', 'guppi_tf_c8_ov16', 'guppi_cf_c6_p1', 'guppi_real_c1'])
@pytest.mark.parametrize('item', [(), 0, -1, slice(3, 17), slice(5, None, 4),
                                  (slice(2, 9), 0), (4, slice(None), 0)])
def test_guppi_payload_items(manifest, name, item):
    from baseband_amd import guppi
    case = manifest[name]
    raw = load_file(case['file'])
    h, hn = orc.guppi_parse_header(raw)
    g = orc.guppi_geometry(h)
    full = orc.guppi_payload_data(raw[hn:hn + g['payload_nbytes']], g)
    with guppi.open(golden_path(case['file']), 'rb') as fb:
        frame = fb.read_frame()
    assert frame.shape == full.shape
    got = frame.payload[item].cpu().numpy()
    want = full[item]
    assert got.shape == np.shape(want)
    assert bits_equal(np.ascontiguousarray(got).reshape(-1), np.ascontiguousarray(want).reshape(-1))
    if item == ():
        assert bits_equal(frame.data.cpu().numpy(), np.ascontiguousarray(full))


@pytest.mark.parametrize('stage', [1, 0])
def test_tiled_kernel_raw_layouts(stage):
    """bb_decode_i8_tiled vs NumPy transposes for all three layouts, odd
    channel counts, partial time ranges (crossing MKBF heaps), several frames,
    payloads at 2-byte (not 4-byte) aligned offsets; `stage` selects
    k_decode_i8_stage (default) or k_decode_i8_tiled for MKBF / time-first."""
    import torch
    from baseband_amd import kernels, _lib
    kernels.tune(_lib.TUNE_TILED_STAGE, stage)
    rng = np.random.default_rng(3)
    for layout, npol, nchan, T, head in ((0, 2, 64, 300, 16), (0, 1, 5, 77, 16), (0, 2, 200, 40, 16),
                                         (1, 2, 32, 512, 16), (1, 1, 3, 256, 16), (2, 2, 8, 100, 16),
                                         (2, 4, 7, 33, 16), (0, 2, 1, 50, 16),
                                         (1, 2, 70, 768, 18), (1, 2, 64, 1024, 16), (1, 4, 9, 512, 6),
                                         (2, 2, 100, 130, 18), (2, 2, 64, 257, 16), (2, 1, 33, 65, 2),
                                         (2, 2, 1024, 20, 16), (1, 2, 1, 256, 16), (2, 2, 1, 64, 16),
                                         (1, 1, 2, 512, 4), (2, 3, 5, 40, 2)):
        nfr = 3
        pn = T * npol * nchan * 2
        raw = rng.integers(0, 256, size=(nfr, pn + head), dtype=np.uint8)
        b = raw[:, head:].view(np.int8)
        if layout == 0:
            ref = b.reshape(nfr, nchan, T, npol, 2).transpose(0, 2, 3, 1, 4)
        elif layout == 1:
            ref = b.reshape(nfr, T // 256, npol, nchan, 256, 2).transpose(0, 1, 4, 2, 3, 5) \
                .reshape(nfr, T, npol, nchan, 2)
        else:
            ref = b.reshape(nfr, T, nchan, npol, 2).transpose(0, 1, 3, 2, 4)
        ref = np.ascontiguousarray(ref).astype(np.float32)
        dbuf = kernels.to_device_bytes(raw.reshape(-1))
        for lo, hi in ((0, T), (3, T - 5), (T // 2, T // 2 + 1)):
            out = kernels.decode_i8_tiled(dbuf, nfr, layout, npol, nchan, T, lo, hi,
                                          src0=head, src_stride=pn + head).cpu().numpy()
            want = ref[:, lo:hi].reshape(-1)
            assert bits_equal(out, np.ascontiguousarray(want)), (layout, npol, nchan, T, lo, hi)
    # missing frame -> fill
    src = torch.tensor([16, -1], dtype=torch.int64, device='cuda')
    out = kernels.decode_i8_tiled(dbuf, 2, 0, 2, 1, 50, 0, 50, src=src,
                                  fill_value=3 - 4j).cpu().numpy().reshape(2, -1, 2)
    assert np.all(out[1] == np.array([3., -4.], np.float32))
    kernels.tune(_lib.TUNE_TILED_STAGE, 1)


@pytest.mark.parametrize('name', DADA_CASES)
def test_dada_stream(manifest, name):
    from baseband_amd import dada
    case = manifest[name]
    exp = load_expected(name)
    with dada.open(golden_path(case['file']), 'rs', squeeze=False) as fh:
        assert fh.shape == exp.shape
        assert bits_equal(fh.read().cpu().numpy(), exp)
        n = exp.shape[0]
        for off, cnt in ((0, 3), (1, 250), (255, 2), (n - 7, 7), (n // 2 + 1, min(300, n - n // 2 - 1))):
            cnt = min(cnt, n - off)
            fh.seek(off)
            assert bits_equal(fh.read(cnt).cpu().numpy(), np.ascontiguousarray(exp[off:off + cnt])), (off, cnt)
        for off, cnt, digest in case.get('reads', []):
            fh.seek(off)
            assert _sha(fh.read(cnt)) == digest


def test_dada_known_answers_and_frame(manifest):
    """dada/tests/test_dada.py:180-183 (first rows), :828-841 (MKBF layout)."""
    from baseband_amd import dada
    exp = load_expected('sample_dada')
    with dada.open(golden_path('samples/sample.dada'), 'rb') as fb:
        frame = fb.read_frame()
        assert frame.shape == (16000, 2, 1)
        assert bits_equal(frame[:3].cpu().numpy(), np.ascontiguousarray(exp[:3]))
        assert bits_equal(frame.payload[100:110, 1].cpu().numpy(), np.ascontiguousarray(exp[100:110, 1]))
        assert bits_equal(frame.data.cpu().numpy(), exp)
    expm = load_expected('sample_mkbf_dada')
    # the sample is cut short of the header's FILE_SIZE: build the payload
    # from the bytes that are there (one heap of 256 samples)
    from baseband_amd.dada import DADAHeader, DADAPayload
    with open(golden_path('samples/sample_mkbf.dada'), 'rb') as f:
        header = DADAHeader.fromfile(f)
        words = np.frombuffer(f.read(), np.uint8)
    h = header.copy()
    h.payload_nbytes = len(words)
    pl = DADAPayload(words.view('<u4'), header=h)
    assert type(pl).__name__ == 'MKBFPayload' and pl.shape == (256, 2, 1024)
    assert bits_equal(pl[5:200:3, 1, 10:20].cpu().numpy(),
                      np.ascontiguousarray(expm[5:200:3, 1, 10:20]))
    with dada.open(golden_path('samples/sample_mkbf.dada'), 'rs', subset=(1, slice(100, 104))) as fh:
        fh.seek(250)
        assert bits_equal(fh.read(6).cpu().numpy(), np.ascontiguousarray(expm[250:256, 1, 100:104]))


def test_gsb_streams(manifest):
    from baseband_amd import gsb
    case = manifest['sample_gsb_rawdump']
    exp = load_expected('sample_gsb_rawdump')
    with gsb.open(golden_path(case['timestamp']), 'rs', raw=golden_path(case['file']),
                  samples_per_frame=8192, squeeze=False) as fh:
        assert bits_equal(fh.read().cpu().numpy(), exp)
        fh.seek(8190)
        assert bits_equal(fh.read(5).cpu().numpy(), exp[8190:8195])
    # gsb/tests/test_gsb.py:257-259: first values of the rawdump sample
    case = manifest['sample_gsb_phased']
    exp = load_expected('sample_gsb_phased')
    raw = [[golden_path(f) for f in pol] for pol in case['files']]
    with gsb.open(golden_path(case['timestamp']), 'rs', raw=raw, samples_per_frame=8,
                  squeeze=False) as fh:
        assert bits_equal(fh.read().cpu().numpy(), exp)
        for off, cnt, digest in case['reads']:
            fh.seek(off)
            assert _sha(fh.read(cnt)) == digest
    with gsb.open(golden_path(case['timestamp']), 'rs', raw=raw, samples_per_frame=8,
                  subset=(1, slice(10, 20))) as fh:
        assert bits_equal(fh.read(9).cpu().numpy(), np.ascontiguousarray(exp[:9, 1, 10:20]))


def test_gsb_payload_fromfile(manifest):
    from baseband_amd.gsb import GSBPayload
    case = manifest['sample_gsb_phased']
    exp = load_expected('sample_gsb_phased')
    fhs = [[open(golden_path(f), 'rb') for f in pol] for pol in case['files']]
    pl = GSBPayload.fromfile(fhs, payload_nbytes=4096, sample_shape=(2, 512), bps=8,
                             complex_data=True)
    assert pl.shape == (8, 2, 512)
    assert bits_equal(pl.data.cpu().numpy(), exp[:8])
    assert bits_equal(pl[3:6, 1].cpu().numpy(), np.ascontiguousarray(exp[3:6, 1]))
    for pol in fhs:
        for f in pol:
            f.close()
    with open(golden_path(manifest['sample_gsb_rawdump']['file']), 'rb') as f:
        pl = GSBPayload.fromfile(f, payload_nbytes=4096, sample_shape=(1,), bps=4)
    assert bits_equal(pl.data.cpu().numpy(), load_expected('sample_gsb_rawdump')[:8192])


def test_dada_float32_passthrough_extension(tmp_path):
    """NBIT 32 (BASELINE config 4; NOT in the reference, which raises KeyError(32)):
    float32 samples come back byte-identical to the file, through the stream
    reader (whole file, windows, partial reads) and the payload."""
    import io
    from baseband_amd import dada
    rng = np.random.default_rng(32)
    spf, npol, nchan, nframes = 1000, 2, 3, 5
    h = dada.DADAHeader.fromvalues(time=np.datetime64('2020-02-02T02:02:02'), sample_rate=1e6,
                                   bps=32, complex_data=True, npol=npol, nchan=nchan,
                                   samples_per_frame=spf)
    assert h.payload_nbytes == spf * npol * nchan * 8
    data = (rng.standard_normal((nframes * spf, npol, nchan))
            + 1j * rng.standard_normal((nframes * spf, npol, nchan))).astype(np.complex64)
    data.view(np.uint32)[7] = 0x7fc01234                  # a NaN payload must survive bit for bit
    p = tmp_path / 'f32.dada'
    with open(str(p), 'wb') as fw:
        for k in range(nframes):
            hk = h.copy()
            hk['OBS_OFFSET'] = k * h.payload_nbytes
            hk.tofile(fw)
            fw.write(data[k * spf:(k + 1) * spf].tobytes())
    with dada.open(str(p), 'rs') as fh:
        assert fh.bps == 32 and fh.shape == (nframes * spf, npol, nchan)
        got = fh.read().cpu().numpy()
        assert bits_equal(got, data)
        fh.window_bytes = 30000                            # several windows per read
        fh.seek(777)
        assert bits_equal(fh.read(3000).cpu().numpy(), data[777:3777])
    with dada.open(str(p), 'rb') as fb:
        frame = fb.read_frame()
        assert bits_equal(frame[10:20].cpu().numpy(), data[10:20])
        assert bits_equal(frame.data.cpu().numpy(), data[:spf])


def test_copy_frames_is_byte_identical():
    """bb_copy_frames (k_copy.h; EXTENSION, no reference counterpart -- parity
    unpinned by construction): strided runs come out byte-identical, with
    16-byte and 4-byte alignment, in a looping grid, with NaN payloads; bad
    arguments are refused."""
    import torch
    from baseband_amd import kernels, _lib
    rng = np.random.default_rng(3232)
    raw = rng.integers(0, 256, 5_000_000, dtype=np.uint8)
    dbuf = kernels.to_device_bytes(raw)
    for nframes, n, src0, stride, blocks in [(1, 4096, 0, 0, 0), (7, 160000, 4096, 164096, 0),
                                             (5, 20004, 12, 20020, 0), (3, 1 << 20, 16, (1 << 20) + 4096, 3),
                                             (300, 16400, 64, 16464, 17), (2, 4, 4, 8, 0)]:
        kernels.tune(_lib.TUNE_BLOCKS, blocks)
        try:
            out = kernels.copy_frames(dbuf, nframes, n, src0=src0, src_stride=stride)
        finally:
            kernels.tune(_lib.TUNE_BLOCKS, 0)
        assert 'k_copy_frames' in _lib.last_kernel()
        assert ('16B' in _lib.last_kernel()) == (n % 16 == 0 and src0 % 16 == 0 and stride % 16 == 0)
        exp = np.concatenate([raw[src0 + i * stride:src0 + i * stride + n] for i in range(nframes)])
        assert out.dtype == torch.float32 and out.numel() == nframes * n // 4
        assert np.array_equal(out.cpu().numpy().view(np.uint8), exp)
    # into a caller's tensor, also one that is not 16-byte aligned
    big = torch.zeros(1000 + 40000 // 4, dtype=torch.float32, device='cuda')
    o = big[1:1 + 10000]
    assert kernels.copy_frames(dbuf, 2, 20000, src0=8, src_stride=30000, out=o) is o
    assert np.array_equal(o.cpu().numpy().view(np.uint8), np.concatenate([raw[8:20008], raw[30008:50008]]))
    assert float(big[0]) == 0 and float(big[10001]) == 0
    for args in [(1, 6, 0, 0), (1, 8, 2, 0), (2, 8, 0, 6)]:                # not multiples of 4
        with pytest.raises(_lib.BBError):
            kernels.copy_frames(dbuf, args[0], args[1], src0=args[2], src_stride=args[3])
    with pytest.raises(_lib.BBError):                                       # a run past the end of the buffer
        kernels.copy_frames(dbuf, 3, 2_000_000, src0=0, src_stride=2_000_000)


def test_block_readers_decode_into_out_tensor(manifest):
    """GUPPI (with overlap) and DADA read(out=device tensor): decoded in place."""
    import torch
    from baseband_amd import guppi, dada
    exp = load_expected('sample_puppi')
    with guppi.open(golden_path('samples/sample_puppi.raw'), 'rs') as fh:
        out = torch.zeros(fh.shape, dtype=torch.complex64, device='cuda')
        assert fh.read(out=out) is out and fh.tell() == fh.shape[0]
        assert bits_equal(out.cpu().numpy(), exp.reshape(out.shape))
        # a read that starts inside an overlap region follows the reference's
        # loop (manifest 'reads'); in place and through a temporary must agree
        fh.seek(1000)
        part = torch.zeros((1500,) + fh.sample_shape, dtype=torch.complex64, device='cuda')
        fh.read(out=part)
        fh.seek(1000)
        assert bits_equal(part.cpu().numpy(), fh.read(1500).cpu().numpy())
        fh.seek(1000)
        host = np.zeros((1500,) + fh.sample_shape, np.complex64)
        fh.read(out=host)
        assert bits_equal(part.cpu().numpy(), host)
    exp = load_expected('sample_dada')
    with dada.open(golden_path('samples/sample.dada'), 'rs') as fh:
        out = torch.zeros(fh.shape, dtype=torch.complex64, device='cuda')
        fh.read(out=out)
        assert bits_equal(out.cpu().numpy(), exp.reshape(out.shape))


def test_gsb_file_level_classes(tmp_path):
    """GSBTimeStampIO / GSBFileReader / GSBFileWriter / GSBFrame.fromdata+tofile
    (gsb/base.py:23-143, gsb/frame.py:99-136) on the sample observations."""
    from baseband_amd import gsb
    d = golden_path('samples/gsb/')
    with gsb.open(d + 'sample_gsb_rawdump.timestamp', 'rt') as ft:
        h0 = ft.read_timestamp()
        assert h0.mode == 'rawdump' and abs(ft.get_frame_rate() - 1e8 / 6 / 2 ** 22) < 1e-6 * 4
    exp = load_expected('sample_gsb_rawdump')
    with gsb.open(d + 'sample_gsb_rawdump.dat', 'rb', payload_nbytes=4096, nchan=1, bps=4) as fr:
        p0 = fr.read_payload()
        assert p0.shape == (8192, 1) and bits_equal(p0.data.cpu().numpy(), exp[:8192].reshape(8192, 1))
        p1 = fr.read_payload()
    # write the two payloads and their timestamps again, frame by frame
    ts, raw = str(tmp_path / 'c.timestamp'), str(tmp_path / 'c.dat')
    with gsb.open(ts, 'wt') as fts, gsb.open(raw, 'wb') as fw:
        fts.write_timestamp(h0)
        fw.write_payload(p0.data, bps=4)
        frame = gsb.GSBFrame.fromdata(p1.data, h0, bps=4)
        frame.tofile(fts, fw)
    assert open(raw, 'rb').read() == open(d + 'sample_gsb_rawdump.dat', 'rb').read()[:8192]
    assert open(ts).read().splitlines() == [' '.join(h0.words)] * 2
    # phased payload split over ((L1, L2), (R1, R2))
    names = [[d + 'sample_gsb_phased.Pol-%s%d.dat' % (p, k) for k in (1, 2)] for p in 'LR']
    with gsb.open(d + 'sample_gsb_phased.timestamp', 'rs', raw=names, samples_per_frame=8) as fs:
        data = fs.read(8)
        hp = fs.header0
    frame = gsb.GSBFrame.fromdata(data, hp, bps=8)
    outs = [[open(str(tmp_path / ('q%d%d.dat' % (p, k))), 'wb') for k in range(2)] for p in range(2)]
    with open(str(tmp_path / 'q.timestamp'), 'w') as fts:
        frame.tofile(fts, outs)
    for p in range(2):
        for k in range(2):
            outs[p][k].close()
            n = 8 // 2 * 512 * 2
            assert open(str(tmp_path / ('q%d%d.dat' % (p, k))), 'rb').read() == open(names[p][k], 'rb').read()[:n]


@pytest.mark.parametrize('fmt,name', [('guppi', n) for n in GUPPI_CASES] + [('dada', n) for n in DADA_CASES])
def test_small_reads_row_staging_equals_whole_frame_staging(manifest, fmt, name):
    """A small request that touches a frame for the first time stages only the
    rows it needs (`_row_range_source`); touching the frame again stages and
    caches the whole frame.  Both must give the same samples (the whole-frame
    path is pinned to the reference by the stream tests above)."""
    import baseband_amd
    mod = getattr(baseband_amd, fmt)
    path = golden_path(manifest[name]['file'])
    rng = np.random.default_rng(len(name))
    with mod.open(path, 'rs', squeeze=False) as fh:
        n, spf = fh.shape[0], fh.samples_per_frame
        if spf < 4:
            pytest.skip('frames too short')
        reads = [(int(rng.integers(0, n - 1)), int(rng.integers(1, max(2, min(spf - 1, 300))))) for _ in range(25)]
        reads = [(o, min(c, n - o)) for o, c in reads]
    for off, cnt in reads:
        with mod.open(path, 'rs', squeeze=False) as fh:
            fh.seek(off)
            first = fh.read(cnt).cpu().numpy()          # random access: rows only (when supported)
            fh.seek(max(off - 1, 0))
            if off > 0:
                fh.read(1)                              # a request that ends at `off` ...
            again = fh.read(cnt).cpu().numpy()          # ... makes this one sequential: whole frame
            fh.seek(off)
            third = fh.read(cnt).cpu().numpy()          # served from the cached frame
        assert bits_equal(first, again) and bits_equal(again, third), (off, cnt)


@pytest.mark.parametrize('tile_rows,tile_chans', [(128, 0), (64, 0), (0, 8), (0, 16), (0, 32), (0, 64), (64, 8), (64, 16), (64, 32), (128, 64)])
def test_xpose_kernel_raw_layouts(tile_rows, tile_chans):
    """k_decode_i8_xpose (16-byte aligned input runs, >= 8 channels): all three
    layouts vs NumPy transposes -- ragged channel tiles, narrow blocks (tiles of
    8, 16, 32 channels), partial time ranges (GUPPI overlap, MKBF heaps),
    several frames, time ranges that end inside a 16-byte piece; and the
    geometries it must hand back to k_tiled.h."""
    from baseband_amd import kernels, _lib
    kernels.tune(_lib.TUNE_XPOSE_ROWS, tile_rows)
    kernels.tune(_lib.TUNE_XPOSE_TC, tile_chans)
    rng = np.random.default_rng(31)
    cases = ((0, 2, 64, 1024, 32), (0, 2, 96, 520, 16), (0, 1, 64, 640, 16), (0, 2, 32, 300, 64),
             (0, 2, 200, 136, 16), (0, 4, 64, 260, 16),
             (0, 2, 8, 2048, 16), (0, 2, 12, 1100, 16), (0, 1, 16, 4096, 16), (0, 2, 24, 520, 32),
             (1, 2, 64, 1024, 16), (1, 2, 96, 512, 32), (1, 1, 64, 768, 16), (1, 2, 32, 256, 16),
             (1, 2, 8, 1024, 16), (1, 1, 16, 2048, 16), (1, 2, 20, 512, 16), (1, 1, 10, 1280, 16),
             (2, 2, 64, 300, 16), (2, 2, 100, 130, 32), (2, 2, 1024, 70, 16), (2, 2, 32, 129, 16),
             (2, 2, 8, 3000, 16), (2, 2, 16, 1029, 16), (2, 2, 28, 700, 16))
    for layout, npol, nchan, T, head in cases:
        nfr = 3
        pn = T * npol * nchan * 2
        stride = pn + head + (-(pn + head)) % 16
        raw = rng.integers(0, 256, size=(nfr, stride), dtype=np.uint8)
        b = np.ascontiguousarray(raw[:, head:head + pn]).view(np.int8)
        if layout == 0:
            ref = b.reshape(nfr, nchan, T, npol, 2).transpose(0, 2, 3, 1, 4)
        elif layout == 1:
            ref = b.reshape(nfr, T // 256, npol, nchan, 256, 2).transpose(0, 1, 4, 2, 3, 5) \
                .reshape(nfr, T, npol, nchan, 2)
        else:
            ref = b.reshape(nfr, T, nchan, npol, 2).transpose(0, 1, 3, 2, 4)
        ref = np.ascontiguousarray(ref).astype(np.float32)
        dbuf = kernels.to_device_bytes(raw.reshape(-1))
        unit = 8 // npol if layout == 0 else (8 if layout == 1 else 1)
        for lo, hi in ((0, T), (unit * 2, T - 5), (0, T - T // 3), (unit * 3, unit * 3 + 1)):
            out = kernels.decode_i8_tiled(dbuf, nfr, layout, npol, nchan, T, lo, hi,
                                          src0=head, src_stride=stride).cpu().numpy()
            assert 'k_decode_i8_xpose' in _lib.last_kernel(), (_lib.last_kernel(), layout, npol, nchan, T, lo)
            want = ref[:, lo:hi].reshape(-1)
            assert bits_equal(out, np.ascontiguousarray(want)), (layout, npol, nchan, T, lo, hi)
        # a start that breaks the 16-byte alignment of the input runs goes to the general kernel
        if layout in (0, 1):
            out = kernels.decode_i8_tiled(dbuf, nfr, layout, npol, nchan, T, 3, T,
                                          src0=head, src_stride=stride).cpu().numpy()
            assert 'k_decode_i8_xpose' not in _lib.last_kernel()
            assert bits_equal(out, np.ascontiguousarray(ref[:, 3:].reshape(-1)))
    # the switch
    kernels.tune(_lib.TUNE_XPOSE, 0)
    try:
        kernels.decode_i8_tiled(dbuf, nfr, layout, npol, nchan, T, 0, T, src0=head, src_stride=stride)
        assert 'k_decode_i8_xpose' not in _lib.last_kernel()
    finally:
        kernels.tune(_lib.TUNE_XPOSE, 1)
        kernels.tune(_lib.TUNE_XPOSE_ROWS, 0)
        kernels.tune(_lib.TUNE_XPOSE_TC, 0)
    # blocks of fewer than 8 channels stay with the general kernel unless a selection asks
    kernels.decode_i8_tiled(dbuf, 1, 0, 2, 4, 64, 0, 64, src0=0, src_stride=1024)
    assert 'k_decode_i8_xpose' not in _lib.last_kernel()


@pytest.mark.parametrize('name', ['sample_puppi', 'guppi_cf_c64_ov0', 'guppi_cf_c64_ov32', 'guppi_cf_c6_p1',
                                  'guppi_tf_c8_ov16', 'sample_mkbf_dada'])
def test_guppi_channel_ranges_are_decoded_alone(manifest, name):
    """GUPPI blocks of both storage orders and MKBF heaps: a subset keeping all
    polarisations and a contiguous channel range enters every block at that
    range (`_plan_channel_range`, `nchan_stored`); result == indexing the
    reference's full decode, for whole reads, reads across the overlap, small
    (row-staged) reads and in-place decodes."""
    import torch
    from baseband_amd import guppi, dada
    if name.endswith('dada'):
        guppi = dada            # (same reader interface; MKBF heaps are the DADA flavour of this layout)
    case = manifest[name]
    exp = load_expected(name)
    n, npol, nchan = exp.shape
    ranges = [slice(1, 3), slice(nchan // 2, None), slice(0, 1), slice(nchan - 1, nchan)]
    for sl in ranges:
        for squeeze in (True, False):
            # (squeezing drops unit axes of the unsliced sample shape; the subset
            # is applied to what is left: base/base.py:706-717)
            subset = (sl,) if (squeeze and npol == 1) else (slice(None), sl)
            shaped = exp.reshape((n,) + tuple(s for s in exp.shape[1:] if s > 1)) if squeeze else exp
            want = shaped[(slice(None),) + subset]
            with guppi.open(golden_path(case['file']), 'rs', subset=subset, squeeze=squeeze) as fh:
                assert fh._within_np is not None and fh._chan_lo == sl.indices(nchan)[0], (sl, squeeze)
                assert fh.shape == want.shape
                assert bits_equal(fh.read().cpu().numpy(), np.ascontiguousarray(want))
                spf = fh.samples_per_frame
                for off, cnt in ((spf - 3, 10), (5, 7), (n - 9, 9), (spf + 1, 2 * spf)):
                    if off + cnt > n:
                        continue
                    # (a read that starts inside a later block takes that block's
                    # own overlap rows: compare with the same read without subset,
                    # which the manifest's digests pin to the reference)
                    with guppi.open(golden_path(case['file']), 'rs', squeeze=False) as ref:
                        ref.seek(off)
                        piece = ref.read(cnt).cpu().numpy()[:, :, sl]
                    if squeeze:
                        piece = piece.reshape((cnt,) + want.shape[1:])
                    fh.seek(off)
                    assert bits_equal(fh.read(cnt).cpu().numpy(), np.ascontiguousarray(piece)), (sl, off, cnt)
                out = torch.empty((n - 2,) + want.shape[1:], dtype=torch.complex64, device='cuda')
                fh.seek(1)
                fh.read(out=out)
                assert bits_equal(out.cpu().numpy(), np.ascontiguousarray(want[1:n - 1]))
    # a channel LIST of all polarisations and a range of ONE polarisation are selections
    # (bb_tiled_params.d_chan_map / pol_first): planned when the kept count is even
    with guppi.open(golden_path(case['file']), 'rs', subset=(slice(None), [0, nchan - 1]), squeeze=False) as fh:
        if nchan > 2:
            assert fh._sel is not None and fh._sel[2].tolist() == [0, nchan - 1]
        assert bits_equal(fh.read().cpu().numpy(), np.ascontiguousarray(exp[:, :, [0, nchan - 1]]))
    if npol > 1:
        with guppi.open(golden_path(case['file']), 'rs', subset=(0, slice(1, 3))) as fh:
            assert fh._sel is not None and fh._sel[:2] == (0, 1) and fh._sel[2].tolist() == [1, 2]
            assert bits_equal(fh.read().cpu().numpy(), np.ascontiguousarray(exp[:, 0, 1:3]))
        with guppi.open(golden_path(case['file']), 'rs', subset=(slice(None), [2, 0, 1])) as fh:
            assert fh._sel is None and fh._within_np is None        # odd count: decode, then index
            assert bits_equal(fh.read().cpu().numpy(), np.ascontiguousarray(exp[:, :, [2, 0, 1]]))


@pytest.mark.parametrize('name', ['sample_puppi', 'guppi_cf_c64_ov0', 'guppi_cf_c64_ov32',
                                  'guppi_tf_c8_ov16', 'sample_mkbf_dada'])
def test_block_channel_lists_and_single_polarisations(manifest, name):
    """channel lists with gaps / in any order and single
    polarisations of GUPPI blocks (both storage orders) and MKBF heaps are
    folded into the decode (`_sel`; bb_tiled_params.d_chan_map, pol_first,
    npol_stored) -- through the fast transposing kernel where the payloads are
    16-byte aligned, through whole-block decode + index where they are not.
    Result == indexing the reference's full decode: whole reads, reads across
    block boundaries and the overlap, small reads, in-place decodes."""
    import torch
    from baseband_amd import guppi, dada
    if name.endswith('dada'):
        guppi = dada
    case = manifest[name]
    exp = load_expected(name)
    n, npol, nchan = exp.shape
    rng = np.random.default_rng(nchan)
    lists = [[nchan - 1, 0], sorted(rng.choice(nchan, size=min(nchan, 6) // 2 * 2, replace=False).tolist()),
             rng.choice(nchan, size=max(2, nchan // 3 // 2 * 2), replace=True).tolist()]
    subsets = [(slice(None), cl) for cl in lists]
    if npol == 2:
        subsets += [(0,), (1,), (1, lists[1]), (0, slice(0, nchan, 2))]
    for subset in subsets:
        want = exp[(slice(None),) + subset]
        with guppi.open(golden_path(case['file']), 'rs', subset=subset) as fh:
            # folded into the decode one way or the other (a drawn list may happen
            # to be a plain range, or everything)
            everything = want.shape[1:] == exp.shape[1:] and np.array_equal(want, exp)
            assert fh._sel is not None or fh._within_np is not None or everything, subset
            assert fh.shape == want.shape, (subset, fh.shape, want.shape)
            assert bits_equal(fh.read().cpu().numpy(), np.ascontiguousarray(want)), subset
            spf = fh.samples_per_frame
            for off, cnt in ((spf - 3, 10), (5, 7), (n - 9, 9), (spf + 1, 2 * spf)):
                if off + cnt > n:
                    continue
                with guppi.open(golden_path(case['file']), 'rs', squeeze=False) as ref:
                    ref.seek(off)
                    piece = ref.read(cnt).cpu().numpy()[(slice(None),) + subset]
                fh.seek(off)
                assert bits_equal(fh.read(cnt).cpu().numpy(), np.ascontiguousarray(piece)), (subset, off, cnt)
            out = torch.empty((n - 2,) + want.shape[1:], dtype=torch.complex64, device='cuda')
            fh.seek(1)
            fh.read(out=out)
            assert bits_equal(out.cpu().numpy(), np.ascontiguousarray(want[1:n - 1])), subset


@pytest.mark.parametrize('tile_rows,tile_chans', [(0, 0), (64, 0), (128, 0), (0, 8), (0, 32), (0, 64), (64, 16), (128, 64)])
@pytest.mark.parametrize('layout', [0, 1, 2])
def test_xpose_kernel_channel_lists_and_polarisations(layout, tile_rows, tile_chans):
    """k_decode_i8_xpose (and k_decode_i8_tf_pick, which takes the channel lists
    of time-first blocks) with a SELECTION: out[f, t, p, c] = stored[f, t,
    pol_first + p, chan_map[c]] -- lists with gaps, repeats and any order, more
    channels than a tile, ragged last tiles, one of two polarisations, both at
    once, partial time ranges; and the geometries that must answer KeyError."""
    import torch
    from baseband_amd import kernels, _lib
    kernels.tune(_lib.TUNE_XPOSE_ROWS, tile_rows)
    kernels.tune(_lib.TUNE_XPOSE_TC, tile_chans)
    rng = np.random.default_rng(700 + layout)
    nfr, nps, stored, T, head = 3, 2, 160, 512, 32
    pn = T * nps * stored * 2
    stride = pn + head
    raw = rng.integers(0, 256, size=nfr * stride, dtype=np.uint8)
    b = np.stack([raw[head + f * stride:head + f * stride + pn] for f in range(nfr)]).view(np.int8)
    if layout == 0:
        ref = b.reshape(nfr, stored, T, nps, 2).transpose(0, 2, 3, 1, 4)
    elif layout == 1:
        ref = b.reshape(nfr, T // 256, nps, stored, 256, 2).transpose(0, 1, 4, 2, 3, 5) \
            .reshape(nfr, T, nps, stored, 2)
    else:
        ref = b.reshape(nfr, T, stored, nps, 2).transpose(0, 1, 3, 2, 4)
    ref = np.ascontiguousarray(ref).astype(np.float32)
    dbuf = kernels.to_device_bytes(raw)
    lists = [np.array([159, 0]), np.sort(rng.choice(stored, 66, replace=False)), rng.choice(stored, 130, replace=True),
             np.arange(0, stored, 2), rng.choice(stored, 8, replace=False), rng.choice(stored, 20, replace=False),
             np.arange(40, 72), None]
    try:
        for cl in lists:
            for pf, npk in ((0, 2), (0, 1), (1, 1)):
                if cl is None and npk == 2:
                    continue                        # (nothing selected: the plain decode)
                for lo, hi in ((0, T), (8, T - 8), (256, 512), (16, 17)):
                    cmap = None if cl is None else torch.from_numpy(cl.astype(np.int32)).cuda()
                    nc = stored if cl is None else cl.size
                    out = kernels.decode_i8_tiled(dbuf, nfr, layout, npk, nc, T, lo, hi, src0=head, src_stride=stride,
                                                  nchan_stored=stored, npol_stored=nps, pol_first=pf,
                                                  chan_map=cmap).cpu().numpy()
                    # (channel lists of time-first blocks: whole rows through LDS, k_tfpick.h --
                    # unless a tile width is forced, which keeps round 3's form in k_xpose.h)
                    picked = layout == 2 and cl is not None and tile_chans == 0
                    assert ('k_decode_i8_tf_pick' if picked else 'k_decode_i8_xpose') in _lib.last_kernel()
                    want = ref[:, lo:hi, pf:pf + npk]
                    if cl is not None:
                        want = want[:, :, :, cl]
                    assert bits_equal(out, np.ascontiguousarray(want).reshape(-1)), (layout, pf, npk, lo, hi, cl)
        cmap = torch.tensor([3, 1, 2], dtype=torch.int32, device='cuda')
        with pytest.raises(KeyError):               # odd channel count: no float4 rows
            kernels.decode_i8_tiled(dbuf, nfr, layout, 2, 3, T, 0, T, src0=head, src_stride=stride,
                                    nchan_stored=stored, npol_stored=nps, chan_map=cmap)
        cmap = torch.tensor([3, 1], dtype=torch.int32, device='cuda')
        with pytest.raises(KeyError):               # payloads off the 16-byte grid
            kernels.decode_i8_tiled(dbuf, nfr, layout, 2, 2, T, 0, T, src0=head + 2, src_stride=stride - 2,
                                    nchan_stored=stored, npol_stored=nps, chan_map=cmap)
        with pytest.raises(_lib.BBError):           # a polarisation the payload does not hold
            kernels.decode_i8_tiled(dbuf, nfr, layout, 1, 2, T, 0, T, src0=head, src_stride=stride,
                                    nchan_stored=stored, npol_stored=nps, pol_first=2, chan_map=cmap)
    finally:
        kernels.tune(_lib.TUNE_XPOSE_ROWS, 0)
        kernels.tune(_lib.TUNE_XPOSE_TC, 0)


@pytest.mark.parametrize('layout', [0, 1, 2])
def test_xpose_kernel_channel_ranges(layout):
    """k_decode_i8_xpose entered at a channel of wider payloads (`nchan_stored`):
    strides follow the stored channel count, tiles the kept range."""
    from baseband_amd import kernels, _lib
    rng = np.random.default_rng(900 + layout)
    nfr, npol, stored, T, head = 3, 2, 160, 512, 32
    pn = T * npol * stored * 2
    stride = pn + head
    raw = rng.integers(0, 256, size=nfr * stride, dtype=np.uint8)
    b = np.stack([raw[head + f * stride:head + f * stride + pn] for f in range(nfr)]).view(np.int8)
    if layout == 0:
        ref = b.reshape(nfr, stored, T, npol, 2).transpose(0, 2, 3, 1, 4)
    elif layout == 1:
        ref = b.reshape(nfr, T // 256, npol, stored, 256, 2).transpose(0, 1, 4, 2, 3, 5) \
            .reshape(nfr, T, npol, stored, 2)
    else:
        ref = b.reshape(nfr, T, stored, npol, 2).transpose(0, 1, 3, 2, 4)
    ref = np.ascontiguousarray(ref).astype(np.float32)
    dbuf = kernels.to_device_bytes(raw)
    for c_lo, keep, lo, hi in ((32, 64, 0, T), (0, 96, 8, T - 8), (96, 64, 256, 512), (4, 36, 0, T),
                               (8, 8, 0, T), (100, 16, 8, T - 8), (12, 24, 0, T)):
        skip = kernels.tiled_channel_skip(layout, npol, T, c_lo)
        out = kernels.decode_i8_tiled(dbuf, nfr, layout, npol, keep, T, lo, hi, src0=head + skip,
                                      src_stride=stride, nchan_stored=stored).cpu().numpy()
        assert 'k_decode_i8_xpose' in _lib.last_kernel(), (_lib.last_kernel(), c_lo, keep)
        want = ref[:, lo:hi, :, c_lo:c_lo + keep]
        assert bits_equal(out, np.ascontiguousarray(want).reshape(-1)), (layout, c_lo, keep, lo, hi)
    with pytest.raises(_lib.BBError):          # fewer stored than decoded channels
        kernels.decode_i8_tiled(dbuf, nfr, layout, npol, 64, T, 0, T, src0=head, src_stride=stride,
                                nchan_stored=32)
