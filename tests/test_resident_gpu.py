"""``open()`` on a file image that already lives in HBM, and ``fh.stage()``:
the headline path through the drop-in API (north_star: "frames staged in HBM
... keeping baseband.open()/StreamReader.read() as the drop-in API";
reference semantics: base/base.py:919-969).  Every golden case of every format
is read through a device tensor and through a staged host file and compared
bit for bit with the reference's output."""
import numpy as np
import pytest

from conftest import golden_path, load_expected, load_file, bits_equal

pytestmark = pytest.mark.gpu

VDIF = ['sample_vdif', 'sample_mwa_vdif', 'sample_arochime_vdif', 'sample_bps1_vdif',
        'vdif_cfg2_small', 'vdif_cfg3_small', 'vdif_bps1_c4', 'vdif_bps4_cplx_t2',
        'vdif_bps8_real_c2', 'vdif_bps8_cplx_t4', 'vdif_bps2_t8_c1', 'vdif_legacy_bps2',
        'vdif_bps4_t2_c1', 'vdif_invalid_fill0', 'vdif_invalid_fillm999']
M5B = ['sample_m5b', 'm5b_c16_b2', 'm5b_c8_b1', 'm5b_c4_b2']
M4 = ['sample_m4', 'sample_32track_m4', 'sample_32track_fanout2_m4', 'sample_16track_m4',
      'sample_64track_fanout2_ft_m4', 'm4_t64_f4', 'm4_t32_f4', 'm4_t32_f2', 'm4_t16_f4']
GUPPI = ['sample_puppi', 'guppi_cf_c64_ov0', 'guppi_cf_c64_ov32', 'guppi_tf_c8_ov16',
         'guppi_cf_c6_p1', 'guppi_real_c1']
DADA = ['sample_dada', 'sample_meerkat_dada', 'sample_mkbf_dada', 'dada_p2_c4_cplx',
        'dada_p1_c1_real', 'dada_p2_c3_real']


def _opener(name, case):
    """(module.open, keyword arguments) for a golden case."""
    import baseband_amd as bb
    kw = {'squeeze': False}
    if name in VDIF:
        if 'frame_rate' in case:
            kw['sample_rate'] = case['frame_rate'] * case['samples_per_frame']
        elif case.get('kwargs'):
            kw['sample_rate'] = case['kwargs']['sample_rate']
        if 'fill_value' in case:
            kw['fill_value'] = case['fill_value']
        return bb.vdif.open, kw
    if name in M5B:
        fr = case.get('frame_rate')
        kw.update(sample_rate=fr * case['samples_per_frame'] if fr else case['sample_rate_hz'],
                  kday=case['kday'], nchan=case['nchan'], bps=case['bps'])
        return bb.mark5b.open, kw
    if name in M4:
        if 'frame_rate' in case:
            kw['sample_rate'] = case['frame_rate'] * case['samples_per_frame']
        kw.update(ntrack=case['ntrack'], decade=2010, verify=False)
        return bb.mark4.open, kw
    if name in GUPPI:
        return bb.guppi.open, kw
    return bb.dada.open, kw


def _device_image(rel):
    import torch
    return torch.from_numpy(load_file(rel).copy()).cuda()


@pytest.mark.parametrize('name', VDIF + M5B + M4 + GUPPI + DADA)
def test_read_from_device_tensor(manifest, name):
    case = manifest[name]
    opener, kw = _opener(name, case)
    exp = load_expected(name)
    with opener(_device_image(case['file']), 'rs', **kw) as fh:
        assert fh.shape == tuple(case['shape'])
        got = fh.read()
        assert got.is_cuda and bits_equal(got.cpu().numpy(), exp)
        # partial reads crossing frame boundaries come from the same image
        spf = fh.samples_per_frame
        n = exp.shape[0]
        for start, count in ((1, min(n - 1, spf + 5)), (max(0, n - 7), min(7, n)), (spf // 2, 1)):
            if start + count > n:
                continue
            fh.seek(start)
            assert bits_equal(fh.read(count).cpu().numpy(), exp[start:start + count]), (start, count)


@pytest.mark.parametrize('name', ['sample_vdif', 'vdif_cfg2_small', 'vdif_cfg3_small', 'sample_m5b',
                                  'sample_m4', 'sample_puppi', 'sample_mkbf_dada', 'sample_dada'])
def test_stage_keeps_file_in_hbm(manifest, name):
    case = manifest[name]
    opener, kw = _opener(name, case)
    exp = load_expected(name)
    with opener(golden_path(case['file']), 'rs', **kw) as fh:
        assert fh.stage() is fh
        assert fh._resident_bytes() is not None
        assert bits_equal(fh.read().cpu().numpy(), exp)
        fh.seek(3)
        assert bits_equal(fh.read(11).cpu().numpy(), exp[3:14])
        fh.unstage()
        fh.seek(0)
        assert bits_equal(fh.read().cpu().numpy(), exp)


@pytest.mark.parametrize('name', ['vdif_cfg2_small', 'vdif_cfg3_small', 'sample_m5b', 'm4_t64_f4'])
@pytest.mark.parametrize('resident', [True, False])
def test_out_tensor_is_decoded_in_place(manifest, name, resident):
    """``read(out=<device tensor>)``: whole frames inside the request are
    decoded straight into `out`, a partial first / last frame through a
    temporary -- any start, any length."""
    import torch
    case = manifest[name]
    opener, kw = _opener(name, case)
    kw['squeeze'] = True
    exp = load_expected(name)
    exp = exp.reshape((exp.shape[0],) + tuple(s for s in exp.shape[1:] if s > 1))
    src = _device_image(case['file']) if resident else golden_path(case['file'])
    with opener(src, 'rs', **kw) as fh:
        spf = fh.samples_per_frame
        n = exp.shape[0]
        tdtype = torch.complex64 if exp.dtype == np.complex64 else torch.float32
        for start, count in ((0, n), (0, 2 * spf), (spf, spf), (5, 2 * spf), (spf - 3, spf + 10),
                             (7, n - 7), (spf + 1, 2 * spf - 1), (3, 9)):
            if start + count > n:
                continue
            out = torch.full((count,) + exp.shape[1:], -77., dtype=tdtype, device='cuda')
            fh.seek(start)
            assert fh.read(out=out) is out
            assert fh.tell() == start + count
            assert bits_equal(out.cpu().numpy(), exp[start:start + count]), (start, count)


def test_device_file_handle_protocol():
    """`DeviceFile`: read / seek / tell like a binary file, headers through the
    lazy strided table."""
    import torch
    from baseband_amd.resident import DeviceFile, DeviceImage
    from baseband_amd.base.header import strided_header_words
    raw = load_file('samples/sample.vdif')
    fh = DeviceFile(torch.from_numpy(raw.copy()).cuda())
    assert fh.read(32) == raw[:32].tobytes() and fh.tell() == 32
    fh.seek(-16, 2)
    assert fh.read() == raw[-16:].tobytes()
    fh.seek(5032)
    assert fh.read(8) == raw[5032:5040].tobytes()
    image = fh.host_image()
    assert isinstance(image, DeviceImage) and len(image) == raw.size
    assert np.array_equal(image[100:300], raw[100:300])
    table = strided_header_words(image, 5032, 8)
    want = strided_header_words(raw, 5032, 8)
    assert len(table) == len(want) == 16
    assert np.array_equal(np.asarray(table), np.asarray(want))
    assert np.array_equal(table[3], want[3]) and np.array_equal(table[2:9, 1], want[2:9, 1])
    with pytest.raises(OSError):
        fh.fileno()


def test_file_reader_on_device_image(manifest):
    """'rb' mode on a device tensor: headers, frames and frame sets."""
    from baseband_amd import vdif
    exp = load_expected('sample_vdif')
    with vdif.open(_device_image('samples/sample.vdif'), 'rb') as fb:
        header = fb.read_header()
        assert header['thread_id'] == 1 and header.frame_nbytes == 5032
        fb.seek(0)
        fs = fb.read_frameset()
        assert bits_equal(fs.data.cpu().numpy(), exp[:20000].reshape(20000, 8, 1))
        assert fb.get_thread_ids() == list(range(8))


def test_reads_back_to_back_without_host_syncs(monkeypatch):
    """read() returns once its frames are verified while the decode goes on
    behind the returned tensor (`_resolve_checks` fetches the verdict on a side
    stream): a loop of reads without host syncs, whose results are dropped at
    once so that the next read's output reuses the same arena block, gives what
    the same loop gives with a sync after every read; and a damaged frame
    still raises at the read that holds it."""
    monkeypatch.setenv('BB_ARENA', '1')
    import torch
    from baseband_amd import arena, placement, synth, vdif
    import bb_oracle_np as orc
    monkeypatch.setattr(placement, 'ARENA_MIN_BYTES', 1 << 20)
    nframes, nf = 6000, 500
    image, h0 = synth.random_vdif(99, nframes, payload_nbytes=8000, frame_rate=1000)
    dev = torch.from_numpy(image.copy()).cuda()
    starts = [((k * 7 + 1) * nf) % (nframes - nf) for k in range(10)]

    def loop(sync):
        sums = []
        with vdif.open(dev, 'rs', sample_rate=32e6) as fh:
            for f0 in starts:
                fh.seek(f0 * 32000)
                got = fh.read(nf * 32000)
                sums.append((got.double().sum(), got[::4001].clone()))
                del got
                if sync:
                    torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [(float(a), b.cpu().numpy()) for a, b in sums]

    try:
        fast, slow = loop(False), loop(True)
        for (a, b), (c, d), f0 in zip(fast, slow, starts):
            assert a == c and bits_equal(b, d), f0
        want, _ = orc.vdif_read(image[starts[3] * 8032:(starts[3] + nf) * 8032], frame_rate=1000)
        assert bits_equal(fast[3][1], want.reshape(-1)[::4001])
        # a frame whose header is not a header: verify=True raises at that read, not later
        bad = dev.clone()
        f_bad = starts[5] + 17
        bad[f_bad * 8032:f_bad * 8032 + 16] = 0xff
        with vdif.open(bad, 'rs', sample_rate=32e6, verify=True) as fh:
            fh.seek(starts[4] * 32000)
            fh.read(10 * 32000)                      # (starts[4] .. +10 does not hold the damage)
            fh.seek(starts[5] * 32000)
            with pytest.raises(AssertionError):      # (the reference's header verification)
                fh.read(nf * 32000)
    finally:
        arena.disable()


@pytest.mark.parametrize('nthread', [1, 8])
def test_scan_on_a_side_stream_gives_the_same_reads(monkeypatch, nthread):
    """Round 5: for requests on bytes that are in HBM already the scan / index /
    verification launches go to a side stream (kernels._FrameWindow,
    bb_vdif_read_window's `scan_stream`), four sets of scratch taking turns, so
    that read() k + 1 gets its verdict while decode k still runs.  Same samples
    as with BB_SIDE_SCAN off -- back to back without host syncs, requests of
    different sizes (below the threshold too), results dropped at once -- and a
    damaged frame raises at the read that holds it."""
    import torch
    from baseband_amd import synth, vdif
    from baseband_amd.base import base as bbase
    pn = 8000 if nthread == 1 else 2000
    nsets = 14000 if nthread == 1 else 7000
    image, h0 = synth.random_vdif(4, nsets, nthread=nthread, nchan=1 if nthread == 1 else 4, bps=2,
                                  complex_data=nthread > 1, payload_nbytes=pn, frame_rate=1000,
                                  thread_order=None if nthread == 1 else [1, 3, 5, 7, 0, 2, 4, 6],
                                  invalid=[(2500, 0), (2501, nthread - 1)])
    spf = h0.samples_per_frame
    rate = 1000 * spf
    dev = torch.from_numpy(image.copy()).cuda()
    big = (17 << 20) // (nthread * (pn + 32)) + 3            # frame sets of a request above the 16 MiB threshold
    plan = [(100, big), (2400, big), (50, 40), (big + 900, big + 77), (0, big), (nsets - big - 1, big), (2450, big)]

    def loop(side):
        monkeypatch.setattr(bbase, '_SIDE_SCAN', side)
        outs = []
        with vdif.open(dev, 'rs', sample_rate=rate, squeeze=False) as fh:
            for f0, n in plan:
                fh.seek(f0 * spf)
                got = fh.read(n * spf)
                outs.append((torch.view_as_real(got).double().sum() if got.is_complex() else got.double().sum(),
                             got.reshape(-1)[::997].clone()))
                del got
            used = fh._scan_stream is not None
        torch.cuda.synchronize()
        return [(float(a), b.cpu().numpy()) for a, b in outs], used

    on, used_on = loop(True)
    off, used_off = loop(False)
    assert used_on and not used_off
    for (a, b), (c, d), pl in zip(on, off, plan):
        assert a == c and np.array_equal(b.view(np.uint8), d.view(np.uint8)), pl
    # damage in a frame set that only the request plan[3] holds
    monkeypatch.setattr(bbase, '_SIDE_SCAN', True)
    bad = dev.clone()
    fset = 5000 if nthread == 1 else 2100
    assert [k for k, (f0, n) in enumerate(plan) if f0 <= fset < f0 + n] == [3]
    fb = fset * nthread * (pn + 32)
    bad[fb:fb + 16] = 0xff
    with vdif.open(bad, 'rs', sample_rate=rate, squeeze=False, verify=True) as fh:
        for k, (f0, n) in enumerate(plan[:5]):
            fh.seek(f0 * spf)
            if k == 3:
                with pytest.raises(AssertionError):  # (the reference's header verification)
                    fh.read(n * spf)
            else:
                fh.read(n * spf)


@pytest.mark.parametrize('fmt', ['mark5b', 'mark4'])
def test_side_stream_scan_for_the_single_thread_formats(monkeypatch, fmt):
    import torch
    import baseband_amd as bb
    from baseband_amd import synth
    from baseband_amd.base import base as bbase
    if fmt == 'mark4':
        image, h0 = synth.random_mark4(3, 260, ntrack=64, fanout=4, frame_rate=400)
        frame = 160000
        op, kw = bb.mark4.open, dict(ntrack=64, decade=2010, sample_rate=400 * 80000)
        spf = 80000
    else:
        nfr = 4200
        rng = np.random.default_rng(8)
        words = rng.integers(0, 2 ** 32, (nfr, 2504), dtype=np.uint64).astype(np.uint32)
        from baseband_amd.mark5b.header import frame_header_words
        words[:, :4] = frame_header_words(np.datetime64('2014-06-13T05:30:01'), 6400, 0, nfr)
        image = words.view(np.uint8).reshape(-1)
        frame, spf = 10016, 5000
        op, kw = bb.mark5b.open, dict(kday=56000, nchan=8, bps=2, sample_rate=6400 * 5000)
    dev = torch.from_numpy(np.ascontiguousarray(image)).cuda()
    n_big = (17 << 20) // frame + 2

    def loop(side):
        monkeypatch.setattr(bbase, '_SIDE_SCAN', side)
        outs = []
        with op(dev, 'rs', **kw) as fh:
            total = fh.shape[0] // spf
            for f0 in (0, 7, total - n_big, 3):
                fh.seek(f0 * spf)
                got = fh.read(n_big * spf)
                outs.append((float(got.double().sum()), got.reshape(-1)[::1013].clone()))
                del got
            used = fh._scan_stream is not None
        torch.cuda.synchronize()
        return outs, used

    on, used = loop(True)
    off, _ = loop(False)
    assert used
    for (a, b), (c, d) in zip(on, off):
        assert a == c and torch.equal(b.view(torch.int32), d.view(torch.int32))


def test_side_stream_scan_soak(monkeypatch):
    """400 reads of random sizes and places from a resident image on two readers
    taking turns, no host syncs, results reduced and dropped at once (temporaries
    come and go in torch's allocator between the reads): every read equals the
    direct decode.  tools/stress_side_scan.py (1,500 reads) found what this
    guards against: scratch newly taken from torch's allocator was filled on the
    side stream while work of its previous user was still queued on the caller's."""
    import torch
    from baseband_amd import synth, vdif, kernels, _lib
    from baseband_amd.base import base as bbase
    monkeypatch.setattr(bbase, '_SIDE_SCAN', True)
    nframes = 36000
    image, h0 = synth.random_vdif(21, nframes, payload_nbytes=8000, frame_rate=1000)
    dev = torch.from_numpy(image.copy()).cuda()
    rng = np.random.default_rng(2)
    plan = []
    for k in range(400):
        n = int(rng.integers(200, 5000)) if k % 3 else int(rng.integers(2100, 9000))
        plan.append((int(rng.integers(0, nframes - n)), n, int(rng.integers(0, 2))))
    sums = torch.zeros(len(plan), dtype=torch.float64, device='cuda')
    fhs = [vdif.open(dev, 'rs', sample_rate=32e6) for _ in range(2)]
    try:
        for k, (f0, n, which) in enumerate(plan):
            fhs[which].seek(f0 * 32000)
            got = fhs[which].read(n * 32000)
            sums[k] = got.double().sum()
            del got
        torch.cuda.synchronize()
        assert all(fh._scan_stream is not None for fh in fhs)
    finally:
        for fh in fhs:
            fh.close()
    want = np.zeros(len(plan))
    for k, (f0, n, which) in enumerate(plan):
        ref = kernels.decode_frames(dev, n, 8000, _lib.CODER_VDIF, 2, src0=32 + f0 * 8032, src_stride=8032)
        want[k] = float(ref.double().sum())
        del ref
    bad = np.nonzero(sums.cpu().numpy() != want)[0]
    assert len(bad) == 0, [(int(k), plan[k]) for k in bad[:5]]


def test_the_pre_read_of_a_window_changes_nothing_but_time():
    """Windows of 16-256 MiB are read through once before their decode (bb_touch inside the
    *_read_window calls, BB_TUNE_TOUCH_MIB): the samples are the same with it, without it, and with
    a limit that leaves this window out; bb_touch itself takes any range, aligned or not."""
    import ctypes as C
    import torch
    import bench
    import bb_oracle_np as orc
    from baseband_amd import kernels, _lib, vdif
    dev = torch.device('cuda')
    nframes = 8192                                          # 63 MiB of file
    image, h0 = bench.make_file_image_on_device(nframes, 77, 0, dev)
    rate = bench.FRAME_RATE * bench.SPF
    got = {}
    try:
        for knob in (256, 0, 32):
            kernels.tune(_lib.TUNE_TOUCH_MIB, knob)
            with vdif.open(image, 'rs', sample_rate=rate) as fh:
                fh.seek(5 * bench.SPF)
                got[knob] = fh.read(6000 * bench.SPF)
        torch.cuda.synchronize()
    finally:
        kernels.tune(_lib.TUNE_TOUCH_MIB, -1)
    assert torch.equal(got[256], got[0]) and torch.equal(got[32], got[0])
    want = orc.decode_flat(image.view(torch.uint8).reshape(nframes, bench.FRAME_NBYTES)[5 + 17, 32:].cpu().numpy(), 'vdif', 2)
    assert bits_equal(got[256][17 * bench.SPF:18 * bench.SPF].cpu().numpy(), want)
    flat = image.view(torch.uint8).reshape(-1)
    for lo, n in ((0, flat.numel()), (3, 1000), (16, 16), (5, 7), (0, 0)):
        assert _lib.lib.bb_touch(C.c_void_p(flat.data_ptr() + lo), n, None) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize('fmt,name', [('guppi', 'sample_puppi.raw'), ('dada', 'sample.dada')])
def test_block_readers_queue_a_read_only_pass_that_changes_nothing(monkeypatch, fmt, name):
    """GUPPI / DADA streams in HBM queue `kernels.touch` over the frames of a request in front of
    their decode (base/blockreader.py `read_through`; an int8 decode whose input comes out of the
    memory-side cache runs 20-25 % faster): the samples are the same with and without; requests that
    want less than half of the frames they touch do not queue one."""
    import torch
    import baseband_amd
    from baseband_amd import kernels
    mod = getattr(baseband_amd, fmt)
    raw = torch.from_numpy(np.fromfile(golden_path('samples/' + name), np.uint8)).cuda()
    calls = []
    real = kernels.touch
    monkeypatch.setattr(kernels, 'TOUCH_MIN_BYTES', 0)
    monkeypatch.setattr(kernels, 'touch', lambda dbuf, lo, n: (calls.append((int(lo), int(n))), real(dbuf, lo, n))[1])
    got = {}
    for on in (True, False):
        with mod.open(raw, 'rs') as fh:
            fh.read_through = on
            whole = fh.read()
            fh.seek(fh.samples_per_frame // 4)
            part = fh.read(3 * fh.samples_per_frame // 4)
            n_before = len(calls)
            fh.seek(3)
            few = fh.read(5)
            assert len(calls) == n_before                   # (five samples of a frame: not worth a pass)
            got[on] = (whole, part, few)
        if on:
            assert len(calls) >= 2 and all(n > 0 for _, n in calls)
            n_on = len(calls)
    assert len(calls) == n_on                               # (none with read_through off)
    for a, b in zip(got[True], got[False]):
        assert torch.equal(a, b)
