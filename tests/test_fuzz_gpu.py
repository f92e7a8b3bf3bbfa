"""Seeded differential fuzz of the decode entry points against the CPU oracle.

The parametrised tests elsewhere pick geometries on purpose; this one draws
them: payload sizes that are and are not multiples of the 256-byte load block,
sources with and without an index, odd byte offsets, missing frames, thread
interleaves of every width, channel selections, and launch sizes from one
frame to a few thousand work items (several workgroups per kernel, striped
work order on and off).  Every case is bit-exact against
`bb_oracle_np.decode_flat` laid out as include/bbdecode.h says.
"""
import numpy as np
import pytest

import bb_oracle_np as orc
from conftest import bits_equal

pytestmark = pytest.mark.gpu

CODERS = {'vdif': 0, 'mark5b': 1, 'int': 2}
COMBOS = [('vdif', 1), ('vdif', 2), ('vdif', 4), ('vdif', 8),
          ('mark5b', 1), ('mark5b', 2), ('int', 4), ('int', 8)]


def _expected(raw, src, nframes, nslot, pn, coder, bps, chunk, cplx, fill, within=None):
    E = pn * 8 // bps
    R = E // chunk
    exp = np.empty((nframes, R, nslot, chunk), np.float32)
    if cplx:
        fillrow = np.tile(np.array([fill[0], fill[1]], np.float32), chunk // 2)
    else:
        fillrow = np.full(chunk, fill[0], np.float32)
    for f in range(nframes):
        for s in range(nslot):
            o = int(src[f * nslot + s])
            if o < 0:
                exp[f, :, s, :] = fillrow
            else:
                exp[f, :, s, :] = orc.decode_flat(raw[o:o + pn], coder, bps).reshape(R, chunk)
    if within is not None:
        exp = exp[..., within]
    return exp.reshape(-1)


def _draw(rng):
    coder, bps = COMBOS[rng.integers(len(COMBOS))]
    per_byte = 8 // bps
    # thread slots and floats per thread sample
    nslot = int(rng.choice([1, 1, 1, 2, 3, 4, 8, 16, 64]))
    chunk = int(rng.choice([1, 2, 4, 8, 16, 32, 64, 128]))
    cplx = bool(chunk % 2 == 0 and rng.integers(2))
    # payload: whole words, a whole number of thread samples
    unit = int(np.lcm(4, max(1, chunk // per_byte)))
    kind = rng.integers(4)
    if kind == 0:
        pn = unit * int(rng.integers(1, 40))
    elif kind == 1:
        pn = 256 * int(rng.integers(1, 40))
    elif kind == 2:
        pn = 256 * int(rng.integers(1, 40)) + unit * int(rng.integers(1, 8))
    else:
        pn = int(rng.choice([5000, 8000, 8192, 10000, 32000]))
    pn = max(unit, pn // unit * unit)
    budget = 6_000_000 if nslot <= 8 else 3_000_000          # decoded floats per case
    most = max(1, budget // (pn * per_byte * nslot))
    nframes = int(min(most, rng.choice([1, 2, 5, 37, 256, 700, 3000])))
    return coder, bps, nslot, chunk, cplx, pn, nframes


@pytest.mark.parametrize('seed', range(300))
def test_decode_frames_random_geometry(seed):
    import torch
    from baseband_amd import kernels
    rng = np.random.default_rng(40000 + seed)
    coder, bps, nslot, chunk, cplx, pn, nframes = _draw(rng)
    nrec = nframes * nslot
    indexed = bool(rng.integers(3))
    # fixed-stride sources are whole words apart (include/bbdecode.h); an index
    # may point anywhere
    header = int(rng.choice([0, 16, 32, 4, 10, 1] if indexed else [0, 16, 32, 4]))
    stride = header + pn
    raw = rng.integers(0, 256, nrec * stride + 64, dtype=np.uint8)
    offs = header + stride * np.arange(nrec, dtype=np.int64)
    fill = (float(rng.choice([0., -7.5, 3.25])), float(rng.choice([0., 1.5])))
    kw = dict(chunk=chunk, nslot=nslot, complex_data=cplx, fill_value=fill[0])
    if cplx:
        kw['fill_value'] = complex(*fill)
    if indexed:
        src = offs[rng.permutation(nrec)]
        nmiss = int(rng.integers(0, max(1, nrec // 8) + 1))
        if nmiss:
            src[rng.choice(nrec, size=nmiss, replace=False)] = -1
        kw['src'] = torch.from_numpy(src).cuda()
    else:
        src = offs
        kw.update(src0=header, src_stride=stride)
    stripes = int(rng.choice([-1, 0, 2]))
    from baseband_amd import _lib
    kernels.tune(_lib.TUNE_WORK_STRIPES, stripes)
    try:
        out = kernels.decode_frames(kernels.to_device_bytes(raw), nframes, pn,
                                    CODERS[coder], bps, **kw).cpu().numpy()
    finally:
        kernels.tune(_lib.TUNE_WORK_STRIPES, -1)
    exp = _expected(raw, src, nframes, nslot, pn, coder, bps, chunk, cplx, fill)
    assert out.size == exp.size
    assert bits_equal(out, exp), (coder, bps, nslot, chunk, cplx, pn, nframes, header,
                                  indexed, stripes, _lib.last_kernel())


@pytest.mark.parametrize('seed', range(100))
def test_decode_frames_select_random_geometry(seed):
    import torch
    from baseband_amd import kernels, _lib
    rng = np.random.default_rng(50000 + seed)
    for _ in range(100):
        coder, bps, nslot, chunk, cplx, pn, nframes = _draw(rng)
        if chunk >= 2 and nslot <= 16:
            break
    nrec = nframes * nslot
    header = int(rng.choice([0, 16, 32]))
    stride = header + pn
    raw = rng.integers(0, 256, nrec * stride + 64, dtype=np.uint8)
    src = (header + stride * np.arange(nrec, dtype=np.int64))[rng.permutation(nrec)]
    if nrec > 4:
        src[rng.choice(nrec, size=nrec // 5, replace=False)] = -1
    nsel = int(rng.integers(1, chunk + 1))
    if rng.integers(2):
        # whole complex channels / pairs, ascending (what a reader's subset gives)
        pairs = np.sort(rng.choice(chunk // 2, size=max(1, nsel // 2), replace=False))
        within = (pairs[:, None] * 2 + np.arange(2)).reshape(-1)
    else:
        within = rng.integers(0, chunk, nsel)
    within = within.astype(np.int32)
    fill = (-7.5, 1.5)
    try:
        out = kernels.decode_frames(
            kernels.to_device_bytes(raw), nframes, pn, CODERS[coder], bps, chunk=chunk,
            nslot=nslot, src=torch.from_numpy(src).cuda(), complex_data=cplx,
            fill_value=complex(*fill) if cplx else fill[0],
            within=torch.from_numpy(within).cuda()).cpu().numpy()
    except KeyError:        # BB_ENOTSUP
        pytest.skip("selection does not fit the staging buffer: readers use the general path")
    exp = _expected(raw, src, nframes, nslot, pn, coder, bps, chunk, cplx, fill, within)
    assert bits_equal(out, exp), (coder, bps, nslot, chunk, cplx, pn, nframes, within.tolist(),
                                  _lib.last_kernel())


@pytest.mark.parametrize('seed', range(100))
def test_decode_mark4_random_geometry(seed):
    """bb_decode_mark4: all five bit maps, units of any length, fill prefixes,
    indexed / fixed-stride sources with missing units, odd byte offsets."""
    import json
    import torch
    from conftest import golden_path
    from baseband_amd import kernels, _lib
    rng = np.random.default_rng(60000 + seed)
    with open(golden_path('mark4_bitmaps.json')) as f:
        maps = json.load(f)
    name = sorted(maps)[rng.integers(len(maps))]
    e = maps[name]
    nt = e['ntrack']
    dt = np.dtype(orc.MARK4_DTYPES[nt])
    isz = dt.itemsize
    nwords = int(rng.choice([1, 7, 64, 100, 640, 1000, 4096, 5000, 20000]))
    fill_words = int(rng.choice([0, 0, 1, 160])) if nwords > 160 else 0
    nunits = int(min(max(1, 3_000_000 // (nwords * nt // 2)), rng.choice([1, 3, 20, 300, 2000])))
    indexed = bool(rng.integers(2))
    # (fixed-stride units start on stream-word boundaries; indexed ones anywhere)
    gap = int(rng.choice([0, isz, 3 * isz, 1, 5] if indexed else [0, isz, 3 * isz]))
    stride = nwords * isz + gap
    lead = int(rng.choice([0, isz, 16, 3] if indexed else [0, isz, 16]))
    raw = rng.integers(0, 256, lead + nunits * stride + 64, dtype=np.uint8)
    offs = lead + stride * np.arange(nunits, dtype=np.int64)
    fill = float(rng.choice([0., -9., 2.5]))
    kw = dict(fill_words=fill_words, fill_value=fill)
    if indexed:
        src = offs[rng.permutation(nunits)]
        if nunits > 3:
            src[rng.choice(nunits, size=nunits // 6, replace=False)] = -1
        kw['src'] = torch.from_numpy(src).cuda()
    else:
        src = offs
        kw.update(src0=lead, src_stride=stride)
    # (16- and 32-track units go through the 64-track kernel as super-words
    # when their geometry allows; odd seeds pin the native word size)
    kernels.tune(_lib.TUNE_M4_WIDEN, 1 - seed % 2)
    try:
        out = kernels.decode_mark4(kernels.to_device_bytes(raw), nunits, nt, nwords,
                                   e['sign_bit'], e['mag_bit'], **kw).cpu().numpy()
    finally:
        kernels.tune(_lib.TUNE_M4_WIDEN, 1)
    per = nt // 2
    exp = np.empty((nunits, nwords * per), np.float32)
    for u in range(nunits):
        o = int(src[u])
        if o < 0:
            exp[u] = fill
            continue
        w = raw[o:o + nwords * isz].copy().view(dt)
        exp[u] = np.ascontiguousarray(orc.mark4_decode(w, e['nchan'], e['fanout'],
                                                       e['signature'])).reshape(-1)
        exp[u, :fill_words * per] = fill
    assert bits_equal(out, exp.reshape(-1)), (name, nwords, fill_words, nunits, gap, lead,
                                              'src' in kw, _lib.last_kernel())


def _tiled_reference(b, layout, nfr, npol, nchan, T):
    if layout == 0:         # (chan, time, pol) -> (time, pol, chan)
        ref = b.reshape(nfr, nchan, T, npol, 2).transpose(0, 2, 3, 1, 4)
    elif layout == 1:       # (heap, pol, chan, 256) -> (heap * 256, pol, chan)
        ref = b.reshape(nfr, T // 256, npol, nchan, 256, 2).transpose(0, 1, 4, 2, 3, 5) \
            .reshape(nfr, T, npol, nchan, 2)
    else:                   # (time, chan, pol) -> (time, pol, chan)
        ref = b.reshape(nfr, T, nchan, npol, 2).transpose(0, 1, 3, 2, 4)
    return np.ascontiguousarray(ref).astype(np.float32)


@pytest.mark.parametrize('seed', range(300))
def test_decode_i8_tiled_random_geometry(seed):
    """bb_decode_i8_tiled: the dispatcher chooses between k_decode_i8_xpose and
    the general kernels by alignment, channel count and time range; whatever it
    picks must equal the NumPy transposes of guppi/payload.py:90-102 and
    dada/payload.py:76-79."""
    import torch
    from baseband_amd import kernels, _lib
    rng = np.random.default_rng(70000 + seed)
    layout = int(rng.integers(3))
    npol = int(rng.choice([1, 2, 2, 2, 4]))
    nchan = int(rng.choice([1, 3, 8, 32, 33, 64, 64, 96, 100, 256, 1024]))
    if layout == 1:
        T = 256 * int(rng.integers(1, 5))
    else:
        T = int(rng.choice([1, 33, 64, 100, 130, 256, 520, 1000]))
    while T * npol * nchan * 2 > 1_500_000 and T > 1:
        T = T // 2 if layout != 1 else max(256, T - 256)
        if layout == 1 and T == 256:
            break
    nfr = int(rng.choice([1, 2, 5]))
    pn = T * npol * nchan * 2
    head = int(rng.choice([0, 16, 32, 64, 2, 18, 6]))
    stride = pn + head
    if rng.integers(2):
        stride += (-stride) % 16
    raw = rng.integers(0, 256, size=nfr * stride + 64, dtype=np.uint8)
    offs = head + stride * np.arange(nfr, dtype=np.int64)
    b = np.stack([raw[o:o + pn] for o in offs]).view(np.int8)
    ref = _tiled_reference(b, layout, nfr, npol, nchan, T)
    lo = int(rng.choice([0, 0, 8, 3, T // 2])) if T > 8 else 0
    hi = int(rng.choice([T, T, T - 1, T - 8, lo + 1])) if T > 8 else T
    lo, hi = min(lo, T - 1), max(min(hi, T), min(lo, T - 1) + 1)
    fill = complex(3., -4.)
    kw = dict(fill_value=fill)
    # a channel range of the stored channels (nchan_stored): enter the payloads
    # at channel c_lo and decode nkeep of them
    c_lo, nkeep = 0, nchan
    if nchan > 1 and rng.integers(2):
        nkeep = int(rng.choice([1, 2, 4, 32, 36, 64]))
        nkeep = min(nkeep, nchan - 1)
        c_lo = int(rng.integers(0, nchan - nkeep + 1))
        if rng.integers(2):
            c_lo -= c_lo % 4                    # (16-byte aligned entry: the fast kernel's case)
        kw['nchan_stored'] = nchan
        skip = kernels.tiled_channel_skip(layout, npol, T, c_lo)
        offs = offs + skip
        head += skip
        ref = np.ascontiguousarray(ref[..., c_lo:c_lo + nkeep, :])
    missing = None
    if rng.integers(2):
        src = offs.copy()
        if nfr > 1 and rng.integers(2):
            missing = int(rng.integers(nfr))
            src[missing] = -1
        kw['src'] = torch.from_numpy(src).cuda()
    else:
        kw.update(src0=head, src_stride=stride)
    out = kernels.decode_i8_tiled(kernels.to_device_bytes(raw), nfr, layout, npol, nkeep, T,
                                  lo, hi, **kw).cpu().numpy()
    want = ref[:, lo:hi].copy()
    if missing is not None:
        want[missing] = np.array([fill.real, fill.imag], np.float32)
    assert bits_equal(out, want.reshape(-1)), (layout, npol, nchan, c_lo, nkeep, T, nfr, head, stride,
                                               lo, hi, missing, _lib.last_kernel())
