"""Mark 4 through the drop-in API on the GPU, bit-exact vs the reference."""
import json

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import golden_path, load_expected, load_file, bits_equal

pytestmark = pytest.mark.gpu

SAMPLES = ['sample_m4', 'sample_32track_m4', 'sample_32track_fanout2_m4',
           'sample_16track_m4', 'sample_64track_fanout2_ft_m4']
SYNTH = ['m4_t64_f4', 'm4_t32_f4', 'm4_t32_f2', 'm4_t16_f4']


def _open(case, **kw):
    from baseband_amd import mark4
    if 'frame_rate' in case:
        kw['sample_rate'] = case['frame_rate'] * case['samples_per_frame']
    return mark4.open(golden_path(case['file']), 'rs', ntrack=case['ntrack'],
                      decade=2010, **kw)


@pytest.mark.parametrize('name', SAMPLES + SYNTH)
def test_stream_read_matches_reference(manifest, name):
    case = manifest[name]
    with _open(case, squeeze=False, verify=False) as fh:
        assert fh.shape == tuple(case['shape'])
        got = fh.read().cpu().numpy()
    assert bits_equal(got, load_expected(name))


@pytest.mark.parametrize('widen', [1, 0])
def test_raw_decode_against_bitmaps_oracle(widen):
    """bb_decode_mark4 on random words for all five modes == the oracle's
    restatement of the reference decoders; 16- and 32-track words natively and
    as 64-bit super-words (word counts 1000 and 19840 qualify)."""
    from baseband_amd import kernels, _lib
    kernels.tune(_lib.TUNE_M4_WIDEN, widen)
    with open(golden_path('mark4_bitmaps.json')) as f:
        maps = json.load(f)
    for name, e in maps.items():
        nt = e['ntrack']
        dt = np.dtype(orc.MARK4_DTYPES[nt])
        for nwords in (1, 63, 64, 65, 1000, 19840):
            rng = np.random.default_rng(nt + nwords)
            w = rng.integers(0, 256, size=(nwords, dt.itemsize), dtype=np.uint8).view(dt).ravel()
            pad = np.zeros(8, np.uint8)
            dbuf = kernels.to_device_bytes(np.concatenate([w.view(np.uint8), pad]))
            out = kernels.decode_mark4(dbuf, 1, nt, nwords, e['sign_bit'], e['mag_bit'])
            ref = orc.mark4_decode(w, e['nchan'], e['fanout'], e['signature'])
            assert bits_equal(out.cpu().numpy(), np.ascontiguousarray(ref).reshape(-1)), (name, nwords)
            if widen and nt < 64 and nwords % (64 // nt) == 0:
                assert 'super-words' in _lib.last_kernel()
    kernels.tune(_lib.TUNE_M4_WIDEN, 1)


def test_known_answers_and_fill_prefix(manifest):
    """mark4/tests/test_mark4.py:449-453,757-765: rows 0-639 of a frame are
    fill, rows 640-641 are the first decoded payload rows."""
    case = manifest['sample_m4']
    exp = load_expected('sample_m4')
    with _open(case, fill_value=-9.) as fh:
        d = fh.read(642).cpu().numpy()
        assert np.all(d[:640] == -9.)
        assert d[640:].astype(int).tolist() == [[-1, 1, 1, -3, -3, -3, 1, -1],
                                                [1, 1, -3, 1, 1, -3, -1, -1]]
        # reads straddling the fill prefix of the second frame
        fh.seek(80000 - 5)
        d = fh.read(650).cpu().numpy()
        want = exp[79995:80645].copy()
        want[5:645] = -9.
        assert bits_equal(d, want)
        fh.seek(-3, 2)
        assert bits_equal(fh.read().cpu().numpy(), exp[-3:])


def test_subset_and_partial(manifest):
    case = manifest['sample_32track_fanout2_m4']
    exp = load_expected('sample_32track_fanout2_m4')
    with _open(case, subset=[0, 5, 7]) as fh:
        assert fh.sample_shape == (3,)
        fh.seek(39000)
        got = fh.read(3000).cpu().numpy()
    assert bits_equal(got, np.ascontiguousarray(exp[39000:42000][:, [0, 5, 7]]))


def test_invalid_frame_from_error_flags(manifest):
    case = manifest['m4_t64_f4']
    assert case['invalid'] == [1]
    exp = load_expected('m4_t64_f4')
    spf = case['samples_per_frame']
    assert np.all(exp[spf:2 * spf] == 0.)
    with _open(case, fill_value=1.5, verify=False) as fh:
        got = fh.read().cpu().numpy()
    want = exp.copy()
    want[spf:2 * spf] = 1.5
    for f in (0, 2):
        want[f * spf:f * spf + 640] = 1.5
    assert bits_equal(got, want)


PAYLOAD_ITEMS = [(), 0, -1, slice(None), slice(3, 13), slice(5, 6), slice(1, None, 3),
                 (slice(2, 40), 1), (9, slice(None)), slice(-7, None)]


@pytest.mark.parametrize('name', ['sample_m4', 'sample_16track_m4',
                                  'sample_64track_fanout2_ft_m4', 'sample_32track_fanout2_m4'])
@pytest.mark.parametrize('item', PAYLOAD_ITEMS)
def test_payload_and_frame_items(manifest, name, item):
    """Payload/frame item matrix (mark4/tests/test_mark4.py:366-388,739-765)."""
    from baseband_amd import mark4
    case = manifest[name]
    exp = load_expected(name)
    spf = case['samples_per_frame']
    nfill = 160 * case['fanout']
    with mark4.open(golden_path(case['file']), 'rb', ntrack=case['ntrack'],
                    decade=2010) as fb:
        fb.seek(case['offset0'])
        frame = fb.read_frame()
    assert len(frame) == spf and frame.valid
    pl = frame.payload
    assert pl.shape == (spf - nfill, case['nchan'])
    full = exp[nfill:spf]
    got = pl[item].cpu().numpy()
    want = full[item]
    assert got.shape == np.shape(want)
    assert bits_equal(got.reshape(-1), np.ascontiguousarray(want).reshape(-1))
    fgot = frame[item].cpu().numpy()
    fwant = exp[:spf][item]
    assert fgot.shape == np.shape(fwant)
    assert bits_equal(fgot.reshape(-1), np.ascontiguousarray(fwant).reshape(-1))


def test_frame_slices_across_fill_boundary(manifest):
    from baseband_amd import mark4
    case = manifest['sample_m4']
    exp = load_expected('sample_m4')[:80000]
    with mark4.open(golden_path(case['file']), 'rb', ntrack=64, decade=2010) as fb:
        fb.seek(case['offset0'])
        frame = fb.read_frame()
    frame.fill_value = 7.
    want = exp.copy()
    want[:640] = 7.
    for item in (slice(630, 650), slice(0, 640), slice(600, 700, 7), 639, 640,
                 slice(635, 645, 2), (slice(638, 642), 3)):
        got = frame[item].cpu().numpy()
        assert bits_equal(np.ascontiguousarray(got).reshape(-1),
                          np.ascontiguousarray(want[item]).reshape(-1)), item


def test_unsupported_mode_is_keyerror():
    from baseband_amd.mark4 import Mark4Payload
    pl = Mark4Payload(np.zeros(100, '<u4'), sample_shape=(16,), bps=1, fanout=2)
    with pytest.raises(KeyError):
        pl.data


@pytest.mark.parametrize('ntrack,fanout,nframes', [(64, 4, 40), (32, 2, 70), (16, 4, 90)])
def test_seeded_synthetic_vs_oracle(ntrack, fanout, nframes, tmp_path):
    """Multi-window files with leading junk bytes and error-flag frames."""
    from baseband_amd import mark4, synth
    from baseband_amd.mark4.base import Mark4StreamReader
    word = ntrack // 8
    image, h0 = synth.random_mark4(ntrack + fanout, nframes, ntrack=ntrack,
                                   fanout=fanout, frame_rate=400,
                                   invalid=[3, nframes - 2], lead_bytes=word * 337)
    p = tmp_path / 'synth.m4'
    p.write_bytes(image.tobytes())
    exp, info = orc.mark4_read(image, ntrack, frame_rate=400)
    assert info['offset0'] == word * 337
    old = Mark4StreamReader.window_bytes
    Mark4StreamReader.window_bytes = 1 << 20
    try:
        with mark4.open(str(p), 'rs', ntrack=ntrack, decade=2010,
                        sample_rate=400 * h0.samples_per_frame, verify=False) as fh:
            got = fh.read().cpu().numpy()
        assert bits_equal(got, exp)
        with mark4.open(str(p), 'rs', decade=2010,
                        sample_rate=400 * h0.samples_per_frame) as fh:   # ntrack auto
            fh.seek(h0.samples_per_frame * 7 + 11)
            part = fh.read(h0.samples_per_frame * 9 + 3).cpu().numpy()
        s0 = h0.samples_per_frame * 7 + 11
        assert bits_equal(part, exp[s0:s0 + h0.samples_per_frame * 9 + 3])
    finally:
        Mark4StreamReader.window_bytes = old


def test_longitudinal_header_crc_matches_reference():
    """bb_mark4_header_crc == the reference's crc12 over the
    160 header stream words of every frame (tests/golden/mark4_crc_cases.json),
    for clean files and files with single header bits flipped; through the raw
    kernel (fixed stride and explicit, unaligned offsets) and the reader
    method; decoded samples are the same whether or not the check is run."""
    import os
    import torch
    from conftest import GOLD
    from baseband_amd import kernels, mark4
    with open(os.path.join(GOLD, 'mark4_crc_cases.json')) as f:
        cases = json.load(f)['cases']
    for c in cases:
        raw = load_file(c['file']).copy()
        for byte, bit in c['flips']:
            raw[byte] ^= np.uint8(1 << bit)
        want = [int(b, 16) for b in c['bad']]
        dev = kernels.to_device_bytes(raw)
        got = kernels.mark4_header_crc(dev, c['nframes'], c['ntrack'], first_offset=c['offset0'])
        assert [int(x) & ((1 << 64) - 1) for x in got.cpu().tolist()] == want, c['file']
        # explicit offsets into a copy shifted by one byte (unaligned words)
        shifted = kernels.to_device_bytes(np.concatenate([np.zeros(1, np.uint8), raw]))
        offs = torch.arange(c['nframes'], dtype=torch.int64, device='cuda') * (c['ntrack'] * 2500) \
            + c['offset0'] + 1
        got = kernels.mark4_header_crc(shifted, c['nframes'], c['ntrack'], offsets=offs)
        assert [int(x) & ((1 << 64) - 1) for x in got.cpu().tolist()] == want, c['file']
    # the reader's report, and that it leaves the samples alone
    case = [c for c in cases if c['file'] == 'samples/sample.m4' and not c['flips']][0]
    with mark4.open(golden_path('samples/sample.m4'), 'rs', ntrack=64, decade=2010) as fh:
        before = fh.read().cpu().numpy()
        bad = fh.header_crc_errors()
        assert bad.numel() == case['nframes'] and not bool(bad.any())
        fh.seek(0)
        assert bits_equal(fh.read().cpu().numpy(), before)


def test_raw_decode_with_channel_selection():
    """bb_decode_mark4_select == the oracle's decode indexed by the selection,
    for all five modes, float4 and scalar store paths, fill prefix, missing
    units, unaligned unit offsets."""
    import torch
    from baseband_amd import kernels
    with open(golden_path('mark4_bitmaps.json')) as f:
        maps = json.load(f)
    rng = np.random.default_rng(77)
    for name, e in sorted(maps.items()):
        nt, nchan = e['ntrack'], e['nchan']
        dt = np.dtype(orc.MARK4_DTYPES[nt])
        isz = dt.itemsize
        for nwords, fill_words in ((1, 0), (63, 0), (64, 0), (1000, 0), (20000, 160), (4100, 7)):
            for trial in range(3):
                m = int(rng.integers(1, nchan + 1))
                chans = rng.choice(nchan, size=m, replace=False) if trial else np.sort(
                    rng.choice(nchan, size=m, replace=False))
                nunits = 4
                lead = (0, isz, 3)[trial]
                stride = nwords * isz + (0, isz, 5)[trial]
                raw = rng.integers(0, 256, lead + nunits * stride + 16, dtype=np.uint8)
                src = lead + stride * np.arange(nunits, dtype=np.int64)
                src[2] = -1
                sign, mag = kernels.mark4_select_maps(e['sign_bit'], e['mag_bit'], nchan, chans)
                out = kernels.decode_mark4(kernels.to_device_bytes(raw), nunits, nt, nwords, sign, mag,
                                           fill_words=fill_words, src=torch.from_numpy(src).cuda(),
                                           fill_value=-9., select=True).cpu().numpy()
                per = e['fanout'] * m
                exp = np.empty((nunits, nwords * per), np.float32)
                for u in range(nunits):
                    if src[u] < 0:
                        exp[u] = -9.
                        continue
                    w = raw[src[u]:src[u] + nwords * isz].copy().view(dt)
                    full = orc.mark4_decode(w, nchan, e['fanout'], e['signature'])
                    exp[u] = np.ascontiguousarray(full[:, chans]).reshape(-1)
                    exp[u, :fill_words * per] = -9.
                assert bits_equal(out, exp.reshape(-1)), (name, nwords, fill_words, chans.tolist())


@pytest.mark.parametrize('name', SAMPLES + SYNTH)
def test_channel_subsets_are_folded_into_the_decode(manifest, name):
    """Readers with a channel subset decode only those channels
    (k_decode_mark4_select) and return what indexing the full decode gives."""
    from baseband_amd import _lib
    case = manifest[name]
    exp = load_expected(name)
    nchan = exp.shape[1]
    subsets = [[0], [nchan - 1, 0], slice(1, None, 2), list(range(nchan))[::-1]]
    for subset in subsets:
        want = exp[:, subset]
        with _open(case, subset=subset, verify=False) as fh:
            assert fh._within_np is not None
            assert fh.shape == want.shape
            got = fh.read().cpu().numpy()
            assert 'k_decode_mark4_select' in _lib.last_kernel()
            assert bits_equal(got, np.ascontiguousarray(want)), (name, subset)
            # in-place decode into a caller's tensor, from an odd offset
            import torch
            n = min(70000, exp.shape[0] - 123)
            out = torch.empty((n,) + want.shape[1:], dtype=torch.float32, device='cuda')
            fh.seek(123)
            fh.read(out=out)
            assert bits_equal(out.cpu().numpy(), np.ascontiguousarray(want[123:123 + n]))
