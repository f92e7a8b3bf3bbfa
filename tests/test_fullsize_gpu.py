"""Parity at BASELINE.json's full sizes through size-independent properties.

The inputs are too large for the oracle to decode whole, so at full size the
tests check (a) level census: the number of output samples at each of the
four levels equals the number of 2-bit codes of that value in the payload
bytes, counted independently with integer tensor ops; (b) a few hundred
frames picked at random are compared bit for bit with the oracle; (c) frames
flagged invalid come back as fill and nothing else does.
"""
import os
import sys

import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)


def _code_census(payload_u8):
    """Counts of the four 2-bit codes in a (n, nbytes) uint8 device tensor."""
    import torch
    counts = torch.zeros(4, dtype=torch.int64, device=payload_u8.device)
    step = max(1, (1 << 28) // payload_u8.shape[1])
    for lo in range(0, payload_u8.shape[0], step):
        b = payload_u8[lo:lo + step].to(torch.int16)
        for j in range(4):
            c = (b >> (2 * j)) & 3
            for k in range(4):
                counts[k] += (c == k).sum()
    return counts.cpu().numpy()


def _level_census(out, levels):
    import torch
    res = []
    step = 1 << 30
    for lv in levels:
        n = 0
        for lo in range(0, out.numel(), step):
            n += int((out[lo:lo + step] == float(lv)).sum().item())
        res.append(n)
    return np.array(res)


@pytest.mark.parametrize('gib', [8.0])
def test_cfg1_headline_size(gib):
    """Synthetic 8 GiB single-thread 2-bit VDIF (BASELINE.json configs[1])."""
    import torch
    import bench
    from baseband_amd import kernels, _lib
    dev = torch.device('cuda')
    nframes = int(gib * 2 ** 30) // bench.FRAME_NBYTES
    image, h0 = bench.make_file_image_on_device(nframes, 4242, 0, dev)
    # flag a sprinkling of frames invalid (bit 31 of word 0)
    rng = np.random.default_rng(1)
    bad = np.sort(rng.choice(nframes, size=nframes // 100, replace=False))
    words = image.view(torch.int32).view(nframes, bench.FRAME_NBYTES // 4)
    badt = torch.from_numpy(bad).to(dev)
    words[badt, 0] = words[badt, 0] | torch.tensor(-2 ** 31, dtype=torch.int32, device=dev)
    pattern, mask = h0.invariant_pattern()
    recs = kernels.vdif_scan(image, nframes, bench.FRAME_NBYTES, 32, pattern, mask,
                             h0['seconds'], h0['frame_nr'], bench.FRAME_RATE)
    src = kernels.build_index(recs, nframes, 1, None)
    out = kernels.decode_frames(image, nframes, 8000, _lib.CODER_VDIF, 2, src=src,
                                fill_value=0.)
    torch.cuda.synchronize()
    # (c) invalid frames -> exactly those are -1 in the index and zero in the output
    s = src.cpu().numpy()
    assert np.array_equal(np.nonzero(s < 0)[0], bad)
    assert np.array_equal(s[s >= 0], (np.nonzero(s >= 0)[0] * bench.FRAME_NBYTES + 32))
    frames = out.view(nframes, bench.SPF)
    assert float(frames[badt].abs().max()) == 0.0
    # (a) level census over the valid frames
    payload = image.view(nframes, bench.FRAME_NBYTES)[:, 32:]
    good = torch.ones(nframes, dtype=torch.bool, device=dev)
    good[badt] = False
    levels = orc.code_levels('vdif', 2)
    want = np.zeros(4, np.int64)
    step = 1 << 16
    for lo in range(0, nframes, step):
        sel = good[lo:lo + step]
        want += _code_census(payload[lo:lo + step][sel])
    got = _level_census(out, levels)
    nzero = int((out == 0).sum().item()) if False else len(bad) * bench.SPF
    assert got.sum() + nzero == out.numel()
    assert np.array_equal(got, want)
    # (b) random frames bit for bit against the oracle
    pick = np.sort(rng.choice(np.setdiff1d(np.arange(nframes), bad), size=200, replace=False))
    raw = payload[torch.from_numpy(pick).to(dev)].cpu().numpy()
    dec = frames[torch.from_numpy(pick).to(dev)].cpu().numpy()
    for i in range(len(pick)):
        assert np.array_equal(dec[i].view(np.uint32),
                              orc.decode_flat(raw[i], 'vdif', 2).view(np.uint32)), pick[i]


def test_cfg2_multithread_large():
    """8 threads x 16 channels complex 2-bit, shuffled thread order, 4 GiB."""
    import torch
    from baseband_amd import kernels, _lib, synth
    dev = torch.device('cuda')
    fn, pn, nth = 8032, 8000, 8
    nsets = (4 << 30) // (fn * nth)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    img = torch.randint(0, 2 ** 31 - 1, (nsets * nth * fn // 4,), generator=g, device=dev,
                        dtype=torch.int64).to(torch.int32)
    h0 = __import__('baseband_amd').vdif.VDIFHeader.fromvalues(
        edv=0, bps=2, nchan=16, complex_data=True, payload_nbytes=pn, station='AA',
        time=np.datetime64('2020-01-01T00:00:00'))
    order = [1, 3, 5, 7, 0, 2, 4, 6]
    hw = synth.vdif_frame_headers(h0, 64, list(range(nth)), 1000, order)    # header words pattern
    v = img.view(nsets, nth, fn // 4)
    w = [int(x) for x in h0.words]
    idx = torch.arange(nsets, device=dev, dtype=torch.int64)
    v[:, :, 0] = (w[0] + idx // 1000).to(torch.int32)[:, None]
    v[:, :, 1] = ((w[1] & 0xff000000) + idx % 1000).to(torch.int32)[:, None]
    v[:, :, 2] = w[2] - (1 << 32) if w[2] >= (1 << 31) else w[2]
    tid = torch.tensor(order, device=dev, dtype=torch.int64)
    w3 = (w[3] & 0xfc00ffff) | (tid << 16)
    w3 = torch.where(w3 >= (1 << 31), w3 - (1 << 32), w3).to(torch.int32)
    v[:, :, 3] = w3[None, :]
    v[:, :, 4:8] = 0
    image = img.view(torch.uint8)
    pattern, mask = h0.invariant_pattern()
    recs = kernels.vdif_scan(image, nsets * nth, fn, 32, pattern, mask, h0['seconds'],
                             h0['frame_nr'], 1000)
    slot = kernels.thread_slot_map(list(range(nth)), dev)
    src = kernels.build_index(recs, nsets, nth, slot)
    out = kernels.decode_frames(image, nsets, pn, _lib.CODER_VDIF, 2, chunk=32, nslot=nth,
                                src=src, complex_data=True)
    torch.cuda.synchronize()
    s = src.view(nsets, nth).cpu().numpy()
    pos = np.array([order.index(t) for t in range(nth)])
    assert np.array_equal(s, (np.arange(nsets)[:, None] * nth + pos[None, :]) * fn + 32)
    # census
    payload = image.view(nsets * nth, fn)[:, 32:]
    want = np.zeros(4, np.int64)
    for lo in range(0, nsets * nth, 1 << 16):
        want += _code_census(payload[lo:lo + (1 << 16)])
    got = _level_census(out, orc.code_levels('vdif', 2))
    assert np.array_equal(got, want)
    # random frame sets against the oracle: out[set] is (1000, 8, 16) complex
    rng = np.random.default_rng(3)
    o = out.view(nsets, 1000, nth, 32)
    for fset in rng.choice(nsets, size=40, replace=False):
        for t in (0, 3, 7):
            raw = payload[fset * nth + pos[t]].cpu().numpy()
            want_t = orc.decode_flat(raw, 'vdif', 2).reshape(1000, 32)
            got_t = o[fset, :, t].cpu().numpy()
            assert np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32)), (fset, t)
    # a reader's subset of 2 of the 16 channels folded into the decode (k_decode_pick, round 5)
    # equals the full decode indexed afterwards -- every sample of the 4 GiB, compared on the device
    keep = torch.tensor([6, 7, 24, 25], dtype=torch.int32, device=dev)           # channels 3 and 12 (re, im)
    part = kernels.decode_frames(image, nsets, pn, _lib.CODER_VDIF, 2, chunk=32, nslot=nth, src=src,
                                 complex_data=True, within=keep)
    assert 'k_decode_pick' in _lib.last_kernel(), _lib.last_kernel()
    p4 = part.view(nsets, 1000, nth, 4)
    step = 1 << 12
    for lo in range(0, nsets, step):
        assert torch.equal(p4[lo:lo + step].view(torch.int32), o[lo:lo + step][..., keep.long()].view(torch.int32)), lo
    # ... and with one thread missing in every 1000th frame set, through the fill path
    src2 = src.clone()
    src2.view(nsets, nth)[::1000, 5] = -1
    part = kernels.decode_frames(image, nsets, pn, _lib.CODER_VDIF, 2, chunk=32, nslot=nth, src=src2,
                                 complex_data=True, within=keep, fill_value=-3. + 2.j)
    p4 = part.view(nsets, 1000, nth, 4)
    fillrow = torch.tensor([-3., 2., -3., 2.], device=dev)
    assert bool((p4[::1000, :, 5] == fillrow).all())
    assert torch.equal(p4[1::1000].view(torch.int32), o[1::1000][..., keep.long()].view(torch.int32))
    assert torch.equal(p4[::1000, :, :5].view(torch.int32), o[::1000, :, :5][..., keep.long()].view(torch.int32))


def _random_bytes(nbytes, seed, dev):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    buf = torch.empty(nbytes // 4 + 64, dtype=torch.int32, device=dev)
    for lo in range(0, buf.numel(), 1 << 28):
        hi = min(buf.numel(), lo + (1 << 28))
        buf[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g, device=dev,
                                   dtype=torch.int64).to(torch.int32)
    return buf.view(torch.uint8)


def test_cfg4_mark5b_16ch_large():
    """BASELINE configs[3], Mark 5B half: 2 GiB of 10016-byte frames, 16
    channels x 2 bit.  Header scan (sync word, BCD time, fill-pattern frames
    -> invalid), census of the sign/magnitude levels, random frames vs oracle."""
    import torch
    from baseband_amd import kernels, _lib
    from baseband_amd.mark5b.header import frame_header_words
    dev = torch.device('cuda')
    fn, pn = 10016, 10000
    nfr = (2 << 30) // fn
    image = _random_bytes(nfr * fn, 11, dev)[:nfr * fn]
    frame_rate = 6400
    t0 = np.datetime64('2014-06-16T05:56:07', 'ns')
    hw = frame_header_words(t0, float(frame_rate), 0, nfr)
    image.view(nfr, fn)[:, :16] = torch.from_numpy(hw.view(np.uint8).reshape(nfr, 16).copy()).to(dev)
    rng = np.random.default_rng(5)
    bad = np.sort(rng.choice(nfr, size=50, replace=False))
    badt = torch.from_numpy(bad).to(dev)
    image.view(nfr, fn)[badt, 16:] = torch.tensor([0x44, 0x33, 0x22, 0x11], dtype=torch.uint8, device=dev).repeat(pn // 4)
    w2 = int(hw[0, 2])
    jday = ((w2 >> 28) & 0xf) * 100 + ((w2 >> 24) & 0xf) * 10 + ((w2 >> 20) & 0xf)
    secs = sum(((w2 >> (4 * i)) & 0xf) * 10 ** i for i in range(5))
    recs = kernels.mark5b_scan(image, nfr, jday * 86400 + secs, 0, frame_rate)
    f = kernels.recs_fields(recs)
    assert np.array_equal(f['time_index'], np.arange(nfr))
    assert np.all(f['flags'] & _lib.FRAME_OK)
    assert np.array_equal(np.nonzero(f['flags'] & _lib.FRAME_INVALID)[0], bad)
    src = kernels.build_index(recs, nfr, 1, None)
    out = kernels.decode_frames(image, nfr, pn, _lib.CODER_MARK5B, 2, chunk=16, src=src, fill_value=0.)
    torch.cuda.synchronize()
    frames = out.view(nfr, pn * 4)
    assert float(frames[badt].abs().max()) == 0.0
    good = torch.ones(nfr, dtype=torch.bool, device=dev)
    good[badt] = False
    payload = image.view(nfr, fn)[:, 16:]
    want = np.zeros(4, np.int64)
    for lo in range(0, nfr, 1 << 15):
        want += _code_census(payload[lo:lo + (1 << 15)][good[lo:lo + (1 << 15)]])
    levels = orc.code_levels('mark5b', 2)
    got = _level_census(out, levels)
    order = np.argsort(levels)                      # census by level value
    assert np.array_equal(got, want), (got, want)
    pick = np.sort(rng.choice(np.setdiff1d(np.arange(nfr), bad), size=100, replace=False))
    raw = payload[torch.from_numpy(pick).to(dev)].cpu().numpy()
    dec = frames[torch.from_numpy(pick).to(dev)].cpu().numpy()
    for i in range(len(pick)):
        assert np.array_equal(dec[i].view(np.uint32), orc.decode_flat(raw[i], 'mark5b', 2).view(np.uint32)), pick[i]


def test_cfg4_mark4_64track_fanout4_large():
    """BASELINE configs[3], Mark 4 half: 2 GiB of 160000-byte frames, 64
    tracks, fanout 4 (8 channels): the first 640 samples of every frame are
    fill (header place), census of the four levels over the rest, random
    frames against the oracle's track demultiplexer."""
    import torch
    from baseband_amd import kernels
    from baseband_amd.mark4._bitmaps import BITMAPS
    dev = torch.device('cuda')
    fn, nwords = 160000, 20000
    nfr = (2 << 30) // fn
    image = _random_bytes(nfr * fn, 13, dev)[:nfr * fn]
    m = BITMAPS[(8, 2, 4)]
    fill = -77.0
    out = kernels.decode_mark4(image, nfr, 64, nwords, m['sign_bit'], m['mag_bit'], fill_words=160,
                               src0=0, src_stride=fn, fill_value=fill)
    torch.cuda.synchronize()
    frames = out.view(nfr, nwords * 4, 8)            # (frame, sample, channel)
    assert bool((frames[:, :640] == fill).all())
    assert not bool((frames[:, 640:] == fill).any())
    # census: every (sign, magnitude) bit pair of the payload words -> one sample
    hi = float(orc.code_levels('vdif', 2)[3])
    n_hi = int((frames[:, 640:].abs() == hi).sum().item())
    n_lo = int((frames[:, 640:].abs() == 1.0).sum().item())
    assert n_hi + n_lo == nfr * (nwords * 4 - 640) * 8
    words = image.view(torch.int64).view(nfr, nwords)[:, 160:]
    want_hi = 0
    sb = torch.tensor(list(m['sign_bit']), device=dev, dtype=torch.int64)
    mb = torch.tensor(list(m['mag_bit']), device=dev, dtype=torch.int64)
    for lo in range(0, nfr, 256):
        w = words[lo:lo + 256].reshape(-1, 1)
        s = (w >> sb[None, :]) & 1
        g = (w >> mb[None, :]) & 1
        want_hi += int((s == g).sum().item())       # sign == magnitude -> +-Hi
    assert n_hi == want_hi
    rng = np.random.default_rng(9)
    for f in rng.choice(nfr, size=12, replace=False):
        w = image[f * fn:(f + 1) * fn].cpu().numpy().view('<u8')[160:]
        want = np.ascontiguousarray(orc.mark4_decode(w, 8, 4, None))
        got = frames[f, 640:].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f


@pytest.mark.parametrize('overlap', [0, 512])
def test_cfg5_guppi_128mib_blocks(overlap):
    """BASELINE configs[4]: GUPPI 8-bit, 2 pol, 64 channels, channels first,
    128 MiB blocks, OVERLAP 0 / 512: checksum of every channel (sum and sum of
    squares of re / im are permutation invariants), random rows vs NumPy."""
    import torch
    from baseband_amd import kernels, _lib
    dev = torch.device('cuda')
    npol, nchan, blk = 2, 64, 128 << 20
    T = blk // (npol * nchan * 2)
    nfr = 8
    image = _random_bytes(nfr * blk, 17, dev)[:nfr * blk]
    keep = T - overlap
    out = kernels.decode_i8_tiled(image, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, keep,
                                  src0=0, src_stride=blk)
    assert 'k_decode_i8_xpose' in _lib.last_kernel()
    torch.cuda.synchronize()
    o = out.view(nfr, keep, npol, nchan, 2)
    b = image.view(torch.int8).view(nfr, nchan, T, npol, 2)[:, :, :keep]
    for f in range(nfr):
        got = o[f].to(torch.float64)
        want = b[f].to(torch.float64)
        # per (pol, chan, re/im): sums and sums of squares over time
        assert torch.equal(got.sum(0), want.sum(1).permute(1, 0, 2))
        assert torch.equal((got * got).sum(0), (want * want).sum(1).permute(1, 0, 2))
    rng = np.random.default_rng(21)
    for f, t in zip(rng.integers(0, nfr, 30), rng.integers(0, keep, 30)):
        want = b[f, :, t].permute(1, 0, 2).to(torch.float32)         # (pol, chan, 2)
        assert torch.equal(o[f, t], want), (f, t)
    # the reader's view of the same thing: first rows of block 1 follow the kept rows of block 0
    assert torch.equal(o[1, 0], b[1, :, 0].permute(1, 0, 2).to(torch.float32))


def test_gather_64_threads_large():
    """64 thread slots x 1 channel 2-bit real (chunk 1: every output float
    comes from another payload), 1 GiB: census + random sets vs oracle."""
    import torch
    from baseband_amd import kernels, _lib
    dev = torch.device('cuda')
    pn, fn, nth = 2000, 2032, 64
    nsets = (1 << 30) // (fn * nth)
    image = _random_bytes(nsets * nth * fn, 23, dev)[:nsets * nth * fn]
    rng = np.random.default_rng(2)
    perm = torch.from_numpy(rng.permutation(nth)).to(dev)
    pos = torch.arange(nsets, device=dev, dtype=torch.int64)[:, None] * nth + perm[None, :]
    src = (pos * fn + 32).reshape(-1).contiguous()
    hole = rng.choice(nsets * nth, size=500, replace=False)
    src[torch.from_numpy(hole).to(dev)] = -1
    out = kernels.decode_frames(image, nsets, pn, _lib.CODER_VDIF, 2, chunk=1, nslot=nth, src=src,
                                fill_value=0.)
    assert 'k_decode_gather' in _lib.last_kernel()
    torch.cuda.synchronize()
    payload = image.view(nsets * nth, fn)[:, 32:]
    valid = (src >= 0)
    rows = torch.div(src[valid] - 32, fn, rounding_mode='floor')
    want = np.zeros(4, np.int64)
    for lo in range(0, rows.numel(), 1 << 16):
        want += _code_census(payload[rows[lo:lo + (1 << 16)]])
    got = _level_census(out, orc.code_levels('vdif', 2))
    assert np.array_equal(got, want)
    assert int((out == 0).sum().item()) == 500 * pn * 4
    o = out.view(nsets, pn * 4, nth)
    s = src.view(nsets, nth).cpu().numpy()
    for fset in rng.choice(nsets, size=10, replace=False):
        for t in (0, 31, 63):
            got_t = o[fset, :, t].cpu().numpy()
            if s[fset, t] < 0:
                assert not got_t.any()
                continue
            raw = image[s[fset, t]:s[fset, t] + pn].cpu().numpy()
            assert np.array_equal(got_t.view(np.uint32), orc.decode_flat(raw, 'vdif', 2).view(np.uint32)), (fset, t)


def test_damaged_gigabyte_through_the_located_frame_index():
    """1 GiB of 8-thread VDIF (16,384 frame sets) with a frame, a whole set and forty bytes of a
    payload taken out, opened from the device image and read with the default repair: the
    frame-set rules of the index (tests/golden/refcases/damaged_streams.json pins them on six-set
    files against the reference) at a size where their tensors count -- exactly the frames that
    are gone come back as fill, frames either side of every hole decode as the oracle decodes
    their bytes, every hole is named, and the search plus the read stay within seconds."""
    import time
    import warnings
    import torch
    import bench
    from baseband_amd import vdif
    dev = torch.device('cuda')
    nthread, nsets = 8, 16384
    fn = bench.FRAME_NBYTES
    image, h0 = bench.make_file_image_on_device(nsets, 99, 0, dev, nthread=nthread, order=tuple(range(nthread)))
    flat = image.view(torch.uint8).reshape(-1)
    keep = torch.ones(flat.numel(), dtype=torch.bool, device=dev)
    one = (5000 * nthread + 3) * fn                       # thread 3 of set 5000
    keep[one:one + fn] = False
    keep[9000 * nthread * fn:9001 * nthread * fn] = False  # all of set 9000
    cut = (12000 * nthread + 5) * fn + 4000               # forty bytes inside the payload of thread 5, set 12000
    keep[cut:cut + 40] = False
    damaged = flat[keep].contiguous()
    del keep
    rate = bench.FRAME_RATE * bench.SPF
    t0 = time.perf_counter()
    with vdif.open(damaged, 'rs', sample_rate=rate, squeeze=False) as fh:
        assert fh.shape[0] == nsets * bench.SPF
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            out = fh.read()
        torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 30.
    said = [str(c.message) for c in caught]
    assert any('frame set 5000. Thread(s) [3] missing' in s for s in said), said
    assert any('frame set 9000. The frame set seems to be missing altogether' in s for s in said), said
    assert any('frame set 12000.' in s and '[5]' in s for s in said), said
    sets = out.reshape(nsets, bench.SPF, nthread)
    gone = {(5000, 3), (12000, 5)} | {(9000, t) for t in range(nthread)}
    for k, t in gone:
        assert float(sets[k, :, t].abs().max()) == 0.0, (k, t)
    # nothing else is fill: every other (set, thread) of the damaged sets and their neighbours
    # equals the oracle's decode of its payload in the intact image
    frames = flat.view(nsets, nthread, fn)
    for k in (4999, 5000, 5001, 8999, 9001, 11999, 12000, 12001, nsets - 1):
        for t in range(nthread):
            if (k, t) in gone:
                continue
            want = orc.decode_flat(frames[k, t, 32:].cpu().numpy(), 'vdif', 2)
            got = sets[k, :, t].cpu().numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, t)
    # a census over everything: exactly the eleven missing frames are zero
    assert int((out == 0).sum().item()) == len(gone) * bench.SPF
