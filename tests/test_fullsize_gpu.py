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
