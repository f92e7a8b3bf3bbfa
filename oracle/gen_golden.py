"""Generate golden vectors from the REAL reference (development container only).

Run with the interpreter that can import the reference (astropy present):

    /opt/conda/bin/python3.9 -W ignore oracle/gen_golden.py

It imports mhvk/baseband from /root/reference (read-only), and
  1. dumps the reference's level / look-up tables as uint32 bit patterns,
  2. copies the reference's small sample DATA files (recordings, not code) to
     tests/golden/samples/ and stores what the reference decodes from them,
  3. writes small seeded synthetic files with the reference's own writers and
     stores file bytes + reference-decoded output,
  4. checks oracle/bb_oracle_np.py against the reference on all of the above
     (asserts; this is the "oracle pinned" step).
Only data (inputs and expected outputs) is written; no reference source text.
"""
import os
import sys
import io
import json
import hashlib
import shutil
import tempfile

import numpy as np

np.asscalar = getattr(np, 'asscalar', lambda a: a.item())
np.alen = getattr(np, 'alen', len)
sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import astropy.units as u                       # noqa: E402
from astropy.time import Time                   # noqa: E402
import baseband                                 # noqa: E402
from baseband import vdif, mark5b, mark4, dada, guppi, gsb   # noqa: E402
from baseband.data import (                     # noqa: E402
    SAMPLE_VDIF, SAMPLE_MWA_VDIF, SAMPLE_AROCHIME_VDIF, SAMPLE_BPS1_VDIF,
    SAMPLE_MARK5B, SAMPLE_MARK4, SAMPLE_MARK4_32TRACK,
    SAMPLE_MARK4_32TRACK_FANOUT2, SAMPLE_MARK4_16TRACK,
    SAMPLE_MARK4_64TRACK_FANOUT2_FT, SAMPLE_DADA, SAMPLE_MEERKAT_DADA,
    SAMPLE_MKBF_DADA, SAMPLE_PUPPI, SAMPLE_GSB_RAWDUMP,
    SAMPLE_GSB_RAWDUMP_HEADER, SAMPLE_GSB_PHASED, SAMPLE_GSB_PHASED_HEADER)
import bb_oracle_np as orc                      # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
SAMPLES = os.path.join(GOLD, 'samples')
EXPECT = os.path.join(GOLD, 'expected')
SYNTH = os.path.join(GOLD, 'synth')
for d in (GOLD, SAMPLES, EXPECT, SYNTH):
    os.makedirs(d, exist_ok=True)

manifest = {'reference': 'mhvk/baseband @ 2025-08-08 (/root/reference)',
            'numpy': np.__version__, 'cases': {}}


class KeepBytesIO(io.BytesIO):
    """BytesIO whose content survives the close() done by stream writers."""
    final = None

    def close(self):
        self.final = self.getvalue()
        super().close()

    def value(self):
        return self.final if self.closed else self.getvalue()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def save_expected(name, data, **extra):
    data = np.ascontiguousarray(data)
    np.savez_compressed(os.path.join(EXPECT, name + '.npz'), data=data)
    entry = dict(shape=list(data.shape), dtype=str(data.dtype), sha256=sha(data))
    entry.update(extra)
    manifest['cases'][name] = entry
    return entry


def copy_sample(path, sub=''):
    dst = os.path.join(SAMPLES, sub, os.path.basename(path))
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    shutil.copyfile(path, dst)
    os.chmod(dst, 0o644)
    return os.path.relpath(dst, GOLD)


# ---------------------------------------------------------------- 1. levels
def gen_levels():
    from baseband.base import encoding
    from baseband.vdif import payload as vp
    from baseband.mark5b import payload as m5p
    from baseband.mark4 import payload as m4p
    from baseband.gsb import payload as gp
    allb = np.arange(256, dtype=np.uint8)
    lv = {
        'decoder_levels_1': bits(encoding.decoder_levels[1]).tolist(),
        'decoder_levels_2': bits(encoding.decoder_levels[2]).tolist(),
        'decoder_levels_4': bits(encoding.decoder_levels[4]).tolist(),
        'decode_8bit': bits(encoding.decode_8bit(allb)).tolist(),
        'vdif_lut1bit': bits(vp.lut1bit).tolist(),
        'vdif_lut2bit': bits(vp.lut2bit).tolist(),
        'vdif_lut4bit': bits(vp.lut4bit).tolist(),
        'mark5b_lut1bit': bits(m5p.lut1bit).tolist(),
        'mark5b_lut2bit': bits(m5p.lut2bit).tolist(),
        'mark4_lut1bit': bits(m4p.lut1bit).tolist(),
        'mark4_lut2bit1': bits(m4p.lut2bit1).tolist(),
        'mark4_lut2bit2': bits(m4p.lut2bit2).tolist(),
        'mark4_lut2bit3': bits(m4p.lut2bit3).tolist(),
        'gsb_decode_4bit': bits(gp.decode_4bit(allb.view(np.int8))
                                .reshape(256, 2)).tolist(),
        'gsb_decode_8bit': bits(gp.decode_8bit(allb.view(np.int8))).tolist(),
    }
    with open(os.path.join(GOLD, 'levels.json'), 'w') as f:
        json.dump(lv, f)
    # pin the oracle
    assert np.array_equal(bits(orc.byte_lut('vdif', 1)), bits(vp.lut1bit))
    assert np.array_equal(bits(orc.byte_lut('vdif', 2)), bits(vp.lut2bit))
    assert np.array_equal(bits(orc.byte_lut('vdif', 4)), bits(vp.lut4bit))
    assert np.array_equal(bits(orc.code_levels('vdif', 8)),
                          bits(encoding.decode_8bit(allb)))
    assert np.array_equal(bits(orc.byte_lut('mark5b', 1)), bits(m5p.lut1bit))
    assert np.array_equal(bits(orc.byte_lut('mark5b', 2)), bits(m5p.lut2bit))
    assert np.array_equal(bits(orc.byte_lut('int', 4)),
                          bits(gp.decode_4bit(allb.view(np.int8)).reshape(256, 2)))
    assert np.array_equal(bits(orc.decode_flat(allb, 'int', 8)),
                          bits(gp.decode_8bit(allb.view(np.int8))))
    print('levels: oracle == reference')


# ------------------------------------------------------- 2. sample files
def gen_vdif_samples():
    cases = [
        ('sample_vdif', SAMPLE_VDIF, {}),
        ('sample_mwa_vdif', SAMPLE_MWA_VDIF, dict(sample_rate=1.28 * u.MHz)),
        ('sample_arochime_vdif', SAMPLE_AROCHIME_VDIF,
         dict(sample_rate=800 / 2048 * u.MHz)),
        ('sample_bps1_vdif', SAMPLE_BPS1_VDIF, dict(sample_rate=8 * u.MHz)),
    ]
    for name, path, kw in cases:
        rel = copy_sample(path)
        with vdif.open(path, 'rs', squeeze=False, **kw) as fh:
            data = fh.read()
            h0 = fh.header0
            info = dict(
                file=rel, kwargs={k: float(v.to_value(u.Hz)) for k, v in kw.items()},
                header0_words=[int(w) for w in h0.words],
                samples_per_frame=int(fh.samples_per_frame),
                sample_rate_hz=float(fh.sample_rate.to_value(u.Hz)),
                bps=int(fh.bps), complex_data=bool(fh.complex_data),
                thread_ids=[int(t) for t in fh._thread_ids],
                start_time=fh.start_time.isot, stop_time=fh.stop_time.isot,
                edv=(int(h0.edv) if h0.edv is not False else False))
        with vdif.open(path, 'rb') as fb:
            fb.seek(0)
            order = []
            while True:
                try:
                    h = fb.read_header()
                except EOFError:
                    break
                order.append([int(h['thread_id']), int(h['frame_nr']),
                              int(h['seconds']), bool(h['invalid_data'])])
                fb.seek(h.payload_nbytes, 1)
            pat, mask = h0.invariant_pattern()
            info['frame_order'] = order
            info['stream_mask'] = [int(m) for m in mask]
        save_expected(name, data, **info)
        # pin the oracle's reference-as-written loop
        raw = np.fromfile(path, dtype=np.uint8)
        fr = int(round(info['sample_rate_hz'] / info['samples_per_frame']))
        out, oinfo = orc.vdif_read(raw, frame_rate=fr)
        assert out.dtype == data.dtype and out.shape == data.shape, (name, out.shape, data.shape)
        assert np.array_equal(out.view(np.uint32), data.view(np.uint32)), name
        assert oinfo['thread_ids'] == info['thread_ids']
        m = orc.vdif_stream_mask(info['edv'])
        assert m == info['stream_mask'], (name, m, info['stream_mask'])
        print('vdif sample', name, data.shape, data.dtype, 'oracle == reference')


def gen_mark5b_sample():
    rel = copy_sample(SAMPLE_MARK5B)
    with mark5b.open(SAMPLE_MARK5B, 'rs', sample_rate=32 * u.MHz, kday=56000,
                     nchan=8, bps=2, squeeze=False) as fh:
        data = fh.read()
        info = dict(file=rel, nchan=8, bps=2, sample_rate_hz=32e6, kday=56000,
                    header0_words=[int(w) for w in fh.header0.words],
                    samples_per_frame=int(fh.samples_per_frame),
                    start_time=fh.start_time.isot, stop_time=fh.stop_time.isot)
    save_expected('sample_m5b', data, **info)
    raw = np.fromfile(SAMPLE_MARK5B, dtype=np.uint8)
    out, _ = orc.mark5b_read(raw, nchan=8, bps=2, frame_rate=6400)
    assert np.array_equal(out.view(np.uint32), data.view(np.uint32))
    print('mark5b sample', data.shape, 'oracle == reference')


# --------------------------------------------------- 3. synthetic (VDIF)
def write_synth(name, blob, data, **info):
    with open(os.path.join(SYNTH, name + '.bin'), 'wb') as f:
        f.write(blob)
    info['file'] = os.path.join('synth', name + '.bin')
    info['file_sha256'] = hashlib.sha256(blob).hexdigest()
    save_expected(name, data, **info)


def levels_for(bps):
    from baseband.base import encoding
    if bps == 8:
        return encoding.decode_8bit(np.arange(256, dtype=np.uint8))
    return encoding.decoder_levels[bps]


def gen_vdif_synth():
    t0 = Time('2020-01-01T00:00:00', precision=9)
    cases = [
        # name, nthread, nchan, bps, complex, edv, samples_per_frame, nframes, seed
        ('vdif_cfg2_small', 1, 1, 2, False, 0, 32000, 12, 12345),
        ('vdif_cfg3_small', 8, 16, 2, True, 0, 1000, 4, 7),
        ('vdif_bps1_c4', 1, 4, 1, False, 0, 8000, 4, 21),
        ('vdif_bps4_cplx_t2', 2, 2, 4, True, 1, 500, 5, 22),
        ('vdif_bps8_real_c2', 1, 2, 8, False, 1, 1000, 4, 23),
        ('vdif_bps8_cplx_t4', 4, 1, 8, True, 0, 256, 6, 24),
        ('vdif_bps2_t8_c1', 8, 1, 2, False, 3, 20000, 2, 25),
        ('vdif_legacy_bps2', 2, 4, 2, False, False, 2000, 4, 26),
        ('vdif_bps4_t2_c1', 2, 1, 4, False, 0, 64, 8, 27),
    ]
    for (name, nthread, nchan, bps, cplx, edv, spf, nframes, seed) in cases:
        rng = np.random.default_rng(seed)
        lev = levels_for(bps)
        shape = (spf * nframes, nthread, nchan)
        data = lev[rng.integers(0, len(lev), size=shape + ((2,) if cplx else ()))]
        data = data.astype(np.float32)
        if cplx:
            data = data.view(np.complex64)[..., 0]
        sample_rate = spf * 100 * u.Hz          # 100 frames per second
        bio = KeepBytesIO()
        kw = dict(sample_rate=sample_rate, samples_per_frame=spf,
                  nthread=nthread, nchan=nchan, complex_data=cplx, bps=bps,
                  edv=edv, station='AA', time=t0)
        if edv == 3:
            kw.pop('samples_per_frame')
            kw['frame_length'] = 629
        with vdif.open(bio, 'ws', squeeze=False, **kw) as fw:
            fw.write(data)
        blob = bio.value()
        # shuffle thread order on disk within each frameset (order is "not mandated")
        if nthread > 1:
            with vdif.open(io.BytesIO(blob), 'rb') as fb:
                fn = fb.read_header().frame_nbytes
            frames = [blob[i:i + fn] for i in range(0, len(blob), fn)]
            perm = list(range(1, nthread, 2)) + list(range(0, nthread, 2))
            out = []
            for s in range(0, len(frames), nthread):
                out += [frames[s + p] for p in perm]
            blob = b''.join(out)
        with vdif.open(io.BytesIO(blob), 'rs', squeeze=False,
                       sample_rate=sample_rate) as fr:
            back = fr.read()
            h0w = [int(w) for w in fr.header0.words]
            thread_ids = [int(t) for t in fr._thread_ids]
        assert np.array_equal(back.view(np.uint32), data.view(np.uint32)), name
        write_synth(name, blob, back, nthread=nthread, nchan=nchan, bps=bps,
                    complex_data=cplx, edv=edv, samples_per_frame=spf,
                    nframes=nframes, seed=seed, frame_rate=100,
                    header0_words=h0w, thread_ids=thread_ids)
        out, _ = orc.vdif_read(np.frombuffer(blob, np.uint8), frame_rate=100)
        assert np.array_equal(out.view(np.uint32), back.view(np.uint32)), name
        print('vdif synth', name, back.shape, back.dtype, 'oracle == reference')


def gen_vdif_invalid():
    """Frames flagged invalid_data decode to fill_value (base/frame.py:191-199)."""
    t0 = Time('2020-01-01T00:00:00', precision=9)
    rng = np.random.default_rng(31)
    nthread, nchan, spf, nframes = 2, 4, 400, 6
    lev = levels_for(2)
    data = lev[rng.integers(0, 4, size=(spf * nframes, nthread, nchan))].astype(np.float32)
    bio = KeepBytesIO()
    with vdif.open(bio, 'ws', squeeze=False, sample_rate=spf * 100 * u.Hz,
                   samples_per_frame=spf, nthread=nthread, nchan=nchan,
                   bps=2, edv=0, station='AA', time=t0) as fw:
        fw.write(data)
    blob = bytearray(bio.value())
    fn = 32 + spf * nchan * 2 // 8
    bad = [(1, 0), (3, 1), (4, 0), (4, 1)]           # (frameset, thread position)
    for fs_, th in bad:
        o = (fs_ * nthread + th) * fn
        blob[o + 3] |= 0x80                          # invalid_data bit (w0.31)
    blob = bytes(blob)
    for fill in (0.0, -999.0):
        with vdif.open(io.BytesIO(blob), 'rs', squeeze=False,
                       sample_rate=spf * 100 * u.Hz, fill_value=fill) as fr:
            back = fr.read()
        nm = 'vdif_invalid_fill%s' % ('0' if fill == 0 else 'm999')
        write_synth(nm, blob, back, nthread=nthread, nchan=nchan, bps=2,
                    complex_data=False, edv=0, samples_per_frame=spf,
                    nframes=nframes, frame_rate=100, fill_value=fill,
                    invalid=[list(b) for b in bad])
        out, _ = orc.vdif_read(np.frombuffer(blob, np.uint8), frame_rate=100,
                               fill_value=fill)
        assert np.array_equal(out.view(np.uint32), back.view(np.uint32))
    print('vdif invalid: oracle == reference')


def gen_mark5b_synth():
    t0 = Time('2014-06-13T05:30:01', precision=9)
    for name, nchan, bps, nframes, seed in (('m5b_c16_b2', 16, 2, 5, 4),
                                           ('m5b_c8_b1', 8, 1, 3, 41),
                                           ('m5b_c4_b2', 4, 2, 4, 42)):
        rng = np.random.default_rng(seed)
        spf = 10000 * 8 // bps // nchan
        lev = levels_for(bps)
        data = lev[rng.integers(0, len(lev), size=(spf * nframes, nchan))].astype(np.float32)
        bio = KeepBytesIO()
        sample_rate = spf * 400 * u.Hz
        with mark5b.open(bio, 'ws', sample_rate=sample_rate, nchan=nchan, bps=bps,
                         time=t0, squeeze=False) as fw:
            fw.write(data)
        blob = bytearray(bio.value())
        invalid = []
        if name == 'm5b_c16_b2':
            # frame 2 replaced by the fill pattern -> invalid (mark5b/frame.py:62-70)
            o = 2 * 10016 + 16
            blob[o:o + 10000] = np.full(2500, 0x11223344, '<u4').tobytes()
            invalid = [2]
        blob = bytes(blob)
        with mark5b.open(io.BytesIO(blob), 'rs', sample_rate=sample_rate, kday=56000,
                         nchan=nchan, bps=bps, squeeze=False) as fr:
            back = fr.read()
            h0w = [int(w) for w in fr.header0.words]
        write_synth(name, blob, back, nchan=nchan, bps=bps, nframes=nframes,
                    frame_rate=400, samples_per_frame=spf, seed=seed,
                    kday=56000, header0_words=h0w, invalid=invalid)
        out, _ = orc.mark5b_read(np.frombuffer(blob, np.uint8), nchan=nchan,
                                 bps=bps, frame_rate=400)
        assert np.array_equal(out.view(np.uint32), back.view(np.uint32)), name
        print('mark5b synth', name, back.shape, 'oracle == reference')


def gen_mark4_bitmaps():
    """Bit -> (sign|magnitude, fanout sample, channel) maps of the five
    registered decoders, found by pushing single-bit words through the
    REFERENCE decoders (facts about the format, stored as data)."""
    from baseband.mark4 import payload as m4p
    maps = {}
    hi = np.float32(3.316505)
    for key, dec in m4p.Mark4Payload._decoders.items():
        nchan, sig, fanout = key
        ntrack = nchan * 2 * fanout
        dt = np.dtype(orc.MARK4_DTYPES[ntrack])
        zero = dec(np.zeros(1, dt))                 # all bits 0 -> all -Hi
        assert np.all(zero == -hi)
        sbit = -np.ones((fanout, nchan), int)
        mbit = -np.ones((fanout, nchan), int)
        for b in range(ntrack):
            out = dec(np.array([1 << b], dtype=dt))  # (fanout, nchan)
            t, c = np.nonzero(out != -hi)
            assert len(t) == 1
            if out[t[0], c[0]] == np.float32(1.):    # sign set, magnitude 0 -> +1
                sbit[t[0], c[0]] = b
            else:                                    # magnitude only -> -1
                assert out[t[0], c[0]] == np.float32(-1.)
                mbit[t[0], c[0]] = b
        assert sbit.min() >= 0 and mbit.min() >= 0
        name = '%d_%s_%d' % (nchan, 'ft' if sig != 2 else '2', fanout)
        maps[name] = dict(nchan=nchan, fanout=fanout, ntrack=ntrack,
                          signature=(None if sig == 2 else int(sig)),
                          sign_bit=sbit.ravel().tolist(), mag_bit=mbit.ravel().tolist())
        # pin the oracle's restatement of the decoders on random words
        rng = np.random.default_rng(ntrack + fanout)
        w = rng.integers(0, 2 ** 63, 4096, dtype=np.uint64).astype(dt) if ntrack == 64 \
            else rng.integers(0, 2 ** ntrack, 4096, dtype=np.uint64).astype(dt)
        assert np.array_equal(orc.mark4_decode(w, nchan, fanout, None if sig == 2 else sig),
                              dec(w))
    with open(os.path.join(GOLD, 'mark4_bitmaps.json'), 'w') as f:
        json.dump(maps, f, indent=1, sort_keys=True)
    print('mark4 bitmaps:', sorted(maps), 'oracle decoders == reference')


def gen_mark4_samples():
    cases = [('sample_m4', SAMPLE_MARK4, 64, 32e6), ('sample_32track_m4', SAMPLE_MARK4_32TRACK, 32, None),
             ('sample_32track_fanout2_m4', SAMPLE_MARK4_32TRACK_FANOUT2, 32, None),
             ('sample_16track_m4', SAMPLE_MARK4_16TRACK, 16, None),
             ('sample_64track_fanout2_ft_m4', SAMPLE_MARK4_64TRACK_FANOUT2_FT, 64, None)]
    for name, path, ntrack, sr in cases:
        rel = copy_sample(path)
        with mark4.open(path, 'rs', ntrack=ntrack, decade=2010, squeeze=False) as fh:
            data = fh.read()
            h0 = fh.header0
            info = dict(file=rel, ntrack=ntrack, decade=2010,
                        sample_rate_hz=float(fh.sample_rate.to_value(u.Hz)),
                        samples_per_frame=int(fh.samples_per_frame),
                        fanout=int(h0.fanout), nchan=int(h0.nchan), bps=int(h0.bps),
                        offset0=int(fh._raw_offsets[0]),
                        header0_words=np.asarray(h0.words).tolist(),
                        start_time=fh.start_time.isot, stop_time=fh.stop_time.isot)
        save_expected(name, data, **info)
        raw = np.fromfile(path, dtype=np.uint8)
        fr = info['sample_rate_hz'] / info['samples_per_frame']
        out, oinfo = orc.mark4_read(raw, ntrack, frame_rate=int(round(fr)))
        assert oinfo['offset0'] == info['offset0'], (oinfo['offset0'], info['offset0'])
        assert np.array_equal(out.view(np.uint32), data.view(np.uint32)), name
        print('mark4 sample', name, data.shape, 'offset0', info['offset0'], 'oracle == reference')


def gen_mark4_synth():
    t0 = Time('2015-03-02T04:05:06.25', precision=9)
    for name, ntrack, fanout, nchan, nframes, seed in (
            ('m4_t64_f4', 64, 4, 8, 3, 5), ('m4_t32_f4', 32, 4, 4, 3, 51),
            ('m4_t32_f2', 32, 2, 8, 4, 52), ('m4_t16_f4', 16, 4, 2, 3, 53)):
        rng = np.random.default_rng(seed)
        spf = 20000 * fanout
        lev = levels_for(2)
        data = lev[rng.integers(0, 4, size=(spf * nframes, nchan))].astype(np.float32)
        sample_rate = spf * 80 * u.Hz          # 12.5 ms frames
        bio = KeepBytesIO()
        with mark4.open(bio, 'ws', sample_rate=sample_rate, time=t0, ntrack=ntrack,
                        bps=2, fanout=fanout, squeeze=False) as fw:
            fw.write(data)
        blob = bytearray(bio.value())
        invalid = []
        if name == 'm4_t64_f4':
            # set communication_error (header word 1 bit 12) on track 5 of frame 1:
            # stream word 32 + (31 - 12) = 51, bit 5
            o = 1 * ntrack * 2500 + 51 * 8
            blob[o] |= 1 << 5
            invalid = [1]
        blob = bytes(blob)
        with mark4.open(io.BytesIO(blob), 'rs', ntrack=ntrack, decade=2010,
                        sample_rate=sample_rate, squeeze=False, verify=False) as fr:
            back = fr.read()
            h0w = np.asarray(fr.header0.words).tolist()
        write_synth(name, blob, back, ntrack=ntrack, fanout=fanout, nchan=nchan,
                    bps=2, nframes=nframes, frame_rate=80, samples_per_frame=spf,
                    seed=seed, decade=2010, header0_words=h0w, invalid=invalid,
                    start_time=t0.isot)
        out, _ = orc.mark4_read(np.frombuffer(blob, np.uint8), ntrack, frame_rate=80)
        assert np.array_equal(out.view(np.uint32), back.view(np.uint32)), name
        print('mark4 synth', name, back.shape, 'oracle == reference')


def gen_guppi():
    rel = copy_sample(SAMPLE_PUPPI)
    raw = np.fromfile(SAMPLE_PUPPI, dtype=np.uint8)
    with guppi.open(SAMPLE_PUPPI, 'rs', squeeze=False) as fh:
        data = fh.read()
        info = dict(file=rel, samples_per_frame=int(fh.samples_per_frame),
                    overlap=int(fh.header0.overlap),
                    sample_rate_hz=float(fh.sample_rate.to_value(u.Hz)),
                    start_time=fh.start_time.isot, stop_time=fh.stop_time.isot,
                    header_nbytes=int(fh.header0.nbytes),
                    payload_nbytes=int(fh.header0.payload_nbytes),
                    channels_first=bool(fh.header0.channels_first))
        reads = []
        for off, cnt in ((0, 100), (900, 200), (960, 64), (1000, 1000), (3000, 904),
                         (3840, 64), (3850, 10), (1919, 3)):
            fh.seek(off)
            reads.append([off, cnt, sha(fh.read(cnt))])
        info['reads'] = reads
    save_expected('sample_puppi', data, **info)
    out, _ = orc.guppi_read(raw)
    assert np.array_equal(out.view(np.uint32), data.view(np.uint32))
    for off, cnt, digest in reads:
        o, _ = orc.guppi_read(raw, off, cnt)
        assert sha(o) == digest, (off, cnt)
    print('guppi sample', data.shape, 'oracle == reference (incl. partial reads)')

    t0 = Time('2018-01-01T00:00:00', precision=9)
    for name, nchan, npol, spf, overlap, nframes, cf, seed in (
            ('guppi_cf_c64_ov0', 64, 2, 512, 0, 3, True, 61),
            ('guppi_cf_c64_ov32', 64, 2, 512, 32, 3, True, 62),
            ('guppi_tf_c8_ov16', 8, 2, 256, 16, 4, False, 63),
            ('guppi_cf_c6_p1', 6, 1, 128, 8, 3, True, 64),
            ('guppi_real_c1', 1, 2, 1024, 0, 2, True, 65)):
        rng = np.random.default_rng(seed)
        cplx = nchan != 1
        n = (spf - overlap) * nframes + overlap
        shape = (n, npol, nchan)
        if cplx:
            data = (rng.integers(-128, 128, size=shape) + 1j * rng.integers(-128, 128, size=shape)).astype(np.complex64)
        else:
            data = rng.integers(-128, 128, size=shape).astype(np.float32)
        tmpf = os.path.join(tempfile.mkdtemp(), 'f.raw')
        # the stream writer refuses OVERLAP != 0, so write frame by frame:
        # frame i holds stream samples [i*keep, i*keep + spf)
        keep = spf - overlap
        bpcs = nchan * npol * (2 if cplx else 1)          # bytes per complete sample
        header0 = guppi.GUPPIHeader.fromvalues(
            time=t0, sample_rate=1 * u.MHz, samples_per_frame=spf, overlap=overlap,
            npol=npol, nchan=nchan, pktsize=keep * bpcs // 4, bps=8,
            pktfmt=('1SFA' if cf else 'SIMPLE'))
        assert header0.channels_first == cf
        ppf = (header0.payload_nbytes - overlap * bpcs) // header0['PKTSIZE']
        with guppi.open(tmpf, 'wb') as fw:
            for i in range(nframes):
                h = header0.copy()
                h.update(pktidx=header0['PKTIDX'] + i * ppf)
                fw.write_frame(data[i * keep:i * keep + spf], h)
        with open(tmpf, 'rb') as ftmp:
            blob = ftmp.read()
        with guppi.open(tmpf, 'rs', squeeze=False) as fr:
            back = fr.read()
            reads = []
            for off, cnt in ((0, 10), (spf - overlap - 3, 2 * spf), (spf - overlap, overlap + 5),
                             (n - overlap - 2, overlap + 2), (n - 1, 1)):
                cnt = min(cnt, n - off)
                fr.seek(off)
                reads.append([off, cnt, sha(fr.read(cnt))])
            hn = int(fr.header0.nbytes)
        assert np.array_equal(back.view(np.uint32), data.view(np.uint32)), name
        write_synth(name, blob, back, nchan=nchan, npol=npol, samples_per_frame=spf - overlap,
                    overlap=overlap, nframes=nframes, channels_first=cf, seed=seed,
                    header_nbytes=hn, reads=reads)
        out, _ = orc.guppi_read(np.frombuffer(blob, np.uint8))
        assert np.array_equal(out.view(np.uint32), back.view(np.uint32)), name
        for off, cnt, digest in reads:
            o, _ = orc.guppi_read(np.frombuffer(blob, np.uint8), off, cnt)
            assert sha(o) == digest, (name, off, cnt)
        print('guppi synth', name, back.shape, back.dtype, 'oracle == reference')


def gen_dada():
    for name, path in (('sample_dada', SAMPLE_DADA), ('sample_meerkat_dada', SAMPLE_MEERKAT_DADA),
                       ('sample_mkbf_dada', SAMPLE_MKBF_DADA)):
        rel = copy_sample(path)
        with dada.open(path, 'rs', squeeze=False) as fh:
            data = fh.read()
            info = dict(file=rel, samples_per_frame=int(fh.samples_per_frame),
                        sample_rate_hz=float(fh.sample_rate.to_value(u.Hz)),
                        start_time=fh.start_time.isot, stop_time=fh.stop_time.isot,
                        header_nbytes=int(fh.header0.nbytes))
            reads = []
            n = data.shape[0]
            for off, cnt in ((0, 3), (1, 250), (255, 2), (n - 7, 7), (n // 2 + 1, 300)):
                cnt = min(cnt, n - off)
                fh.seek(off)
                reads.append([off, cnt, sha(fh.read(cnt))])
            info['reads'] = reads
        save_expected(name, data, **info)
        out, _ = orc.dada_read(np.fromfile(path, dtype=np.uint8))
        assert np.array_equal(out.view(np.uint32), data.view(np.uint32)), name
        for off, cnt, digest in reads:
            assert sha(out[off:off + cnt]) == digest
        print('dada sample', name, data.shape, data.dtype, 'oracle == reference')
    # synthetic: several frames in one file, last one truncated
    t0 = Time('2018-01-01T00:00:00', precision=9)
    for name, npol, nchan, cplx, spf, nframes, cut, seed in (
            ('dada_p2_c4_cplx', 2, 4, True, 1000, 3, 1234, 71),
            ('dada_p1_c1_real', 1, 1, False, 4096, 2, 0, 72),
            ('dada_p2_c3_real', 2, 3, False, 500, 3, 77, 73)):
        rng = np.random.default_rng(seed)
        shape = (spf * nframes, npol, nchan)
        if cplx:
            data = (rng.integers(-128, 128, size=shape) + 1j * rng.integers(-128, 128, size=shape)).astype(np.complex64)
        else:
            data = rng.integers(-128, 128, size=shape).astype(np.float32)
        tmpf = os.path.join(tempfile.mkdtemp(), 'f.dada')
        header0 = dada.DADAHeader.fromvalues(
            time=t0, sample_rate=1 * u.MHz, samples_per_frame=spf, npol=npol,
            nchan=nchan, bps=8, complex_data=cplx)
        with dada.open(tmpf, 'wb') as fw:
            for i in range(nframes):
                h = header0.copy()
                h.update(offset=i * spf / (1 * u.MHz))
                fw.write_frame(data[i * spf:(i + 1) * spf], h)
        with open(tmpf, 'rb') as ftmp:
            blob = ftmp.read()
        if cut:
            blob = blob[:-cut]
        tmpf = os.path.join(tempfile.mkdtemp(), 'all.dada')
        with open(tmpf, 'wb') as ftmp:
            ftmp.write(blob)
        with dada.open(tmpf, 'rs', squeeze=False) as fr:
            back = fr.read()
        assert np.array_equal(back.view(np.uint32), data[:back.shape[0]].view(np.uint32)), name
        write_synth(name, blob, back, npol=npol, nchan=nchan, complex_data=cplx,
                    samples_per_frame=spf, nframes=nframes, cut=cut, seed=seed)
        out, _ = orc.dada_read(np.frombuffer(blob, np.uint8))
        assert np.array_equal(out.view(np.uint32), back.view(np.uint32)), name
        print('dada synth', name, back.shape, back.dtype, 'oracle == reference')


def gen_gsb():
    rel_ts = copy_sample(SAMPLE_GSB_RAWDUMP_HEADER, 'gsb')
    rel = copy_sample(SAMPLE_GSB_RAWDUMP, 'gsb')
    with gsb.open(SAMPLE_GSB_RAWDUMP_HEADER, 'rs', raw=SAMPLE_GSB_RAWDUMP,
                  samples_per_frame=8192, squeeze=False) as fh:
        data = fh.read()
        info = dict(file=rel, timestamp=rel_ts, samples_per_frame=8192,
                    payload_nbytes=int(fh.payload_nbytes), bps=int(fh.bps),
                    sample_rate_hz=float(fh.sample_rate.to_value(u.Hz)),
                    start_time=fh.start_time.isot, stop_time=fh.stop_time.isot)
    save_expected('sample_gsb_rawdump', data, **info)
    out = orc.gsb_read_rawdump(np.fromfile(SAMPLE_GSB_RAWDUMP, np.uint8), 10, info['payload_nbytes'])
    assert np.array_equal(out.view(np.uint32), data.view(np.uint32))
    print('gsb rawdump', data.shape, 'oracle == reference')
    rel_ts = copy_sample(SAMPLE_GSB_PHASED_HEADER, 'gsb')
    rels = [[copy_sample(f, 'gsb') for f in pol] for pol in SAMPLE_GSB_PHASED]
    with gsb.open(SAMPLE_GSB_PHASED_HEADER, 'rs', raw=SAMPLE_GSB_PHASED,
                  samples_per_frame=8, squeeze=False) as fh:
        data = fh.read()
        info = dict(files=rels, timestamp=rel_ts, samples_per_frame=8,
                    payload_nbytes=int(fh.payload_nbytes), bps=int(fh.bps), nchan=512,
                    sample_rate_hz=float(fh.sample_rate.to_value(u.Hz)),
                    start_time=fh.start_time.isot, stop_time=fh.stop_time.isot)
        reads = []
        for off, cnt in ((0, 5), (3, 9), (7, 2), (70, 10), (37, 1)):
            fh.seek(off)
            reads.append([off, cnt, sha(fh.read(cnt))])
        info['reads'] = reads
    save_expected('sample_gsb_phased', data, **info)
    files = [[np.fromfile(f, np.uint8) for f in pol] for pol in SAMPLE_GSB_PHASED]
    out = orc.gsb_read_phased(files, 10, info['payload_nbytes'])
    assert np.array_equal(out.view(np.uint32), data.view(np.uint32))
    print('gsb phased', data.shape, 'oracle == reference')


def gen_vdif_edv_ab():
    """Mark 5B frames wrapped in VDIF (EDV 0xab, vdif/payload.py:151-154):
    the four frames of sample.m5b converted with the reference's
    VDIFFrame.from_mark5b_frame and read back as a VDIF stream."""
    bio = KeepBytesIO()
    with mark5b.open(SAMPLE_MARK5B, 'rb', kday=56000, nchan=8, bps=2) as fh:
        fh.find_header()
        for i in range(4):
            m5f = fh.read_frame()
            vdif.VDIFFrame.from_mark5b_frame(m5f).tofile(bio)
    blob = bio.getvalue()
    # the reference cannot open such a file as a STREAM (its invariant_pattern
    # trips over the fixed complex_data of VDIFMark5BHeader), so the pin is at
    # frame level: VDIFFrame.fromfile(...).data for every frame
    parts = []
    with vdif.open(io.BytesIO(blob), 'rb') as fb:
        for i in range(4):
            fr = fb.read_frame()
            assert fr.header.edv == 0xab
            if i == 0:
                h0w = [int(w) for w in fr.header.words]
            parts.append(fr.data)
    back = np.concatenate(parts)[:, np.newaxis, :]
    info = dict(nthread=1, nchan=8, bps=2, complex_data=False, edv=0xab,
                samples_per_frame=5000, nframes=4, frame_rate=6400,
                header0_words=h0w, thread_ids=[0], frame_level_only=True)
    write_synth('vdif_edv_ab', blob, back, **info)
    with mark5b.open(SAMPLE_MARK5B, 'rs', sample_rate=32 * u.MHz, kday=56000, nchan=8,
                     bps=2) as f5:
        assert np.array_equal(f5.read(), back[:, 0])
    out, _ = orc.vdif_read(np.frombuffer(blob, np.uint8), frame_rate=6400)
    assert np.array_equal(out.view(np.uint32), back.view(np.uint32))
    print('vdif edv 0xab', back.shape, 'oracle == reference')


def gen_encode():
    """Reference ENCODERS on continuous float32 inputs (values near and far
    from the decision thresholds).  Stored: the inputs and the packed words."""
    from baseband.vdif import payload as vp
    from baseband.mark5b import payload as m5p
    from baseband.gsb import payload as gp
    from baseband.dada import payload as dp
    from baseband.base import encoding as enc
    rng = np.random.default_rng(2024)
    x = (rng.standard_normal(8192) * 2.2).astype(np.float32)
    # exact thresholds, levels, halves, extremes, signed zeros
    special = np.array([0., -0., 2.174564, -2.174564, 4.349128, -4.349128, 3.261846,
                        -3.261846, 1e9, -1e9, 0.5, -0.5, 1.5, 2.5, -1.5, -2.5, 127.5,
                        -128.5, 126.5, 3.316505, -3.316505, 1., -1., 0.16949153,
                        -0.16949153, 7.5 / 2.95, -7.5 / 2.95, 2.5423729, 1e-30, -1e-30,
                        0.014084507, -0.014084507], dtype=np.float32)
    x[:len(special)] = special
    thr = np.float32(2.174564)
    x[64:96] = np.nextafter(thr, np.float32(10)) * np.array([1, -1] * 16, np.float32)
    x[96:128] = np.nextafter(thr, np.float32(-10)) * np.array([1, -1] * 16, np.float32)
    out = {'input': x}
    for name, fn, shape in (('vdif1', vp.encode_1bit, None), ('vdif2', vp.encode_2bit, None),
                            ('vdif4', vp.encode_4bit, None), ('vdif8', enc.encode_8bit, None),
                            ('mark5b1', m5p.encode_1bit, None), ('mark5b2', m5p.encode_2bit, None),
                            ('int4', gp.encode_4bit, None), ('int8', dp.encode_8bit, None)):
        out[name] = np.ascontiguousarray(fn(x.copy())).view(np.uint8).ravel()
    for coder, bps, key in (('vdif', 1, 'vdif1'), ('vdif', 2, 'vdif2'), ('vdif', 4, 'vdif4'),
                            ('vdif', 8, 'vdif8'), ('mark5b', 1, 'mark5b1'), ('mark5b', 2, 'mark5b2'),
                            ('int', 4, 'int4'), ('int', 8, 'int8')):
        got = orc.encode_flat(x, coder, bps) if bps < 8 else orc.encode_codes(x, coder, bps)
        assert np.array_equal(got, out[key]), key
    # Mark 4: the five modes, from (nsample, nchan) data
    from baseband.mark4 import payload as m4p
    for key, encfn in m4p.Mark4Payload._encoders.items():
        nchan, sig, fanout = key
        data = x[:nchan * fanout * (8192 // (nchan * fanout))].reshape(-1, nchan)
        w = encfn(data.copy())
        out['mark4_%d_%s_%d' % (nchan, 'ft' if sig != 2 else '2', fanout)] = \
            np.ascontiguousarray(w).view(np.uint8).ravel()
    np.savez_compressed(os.path.join(GOLD, 'encode_cases.npz'), **out)
    print('encode: oracle == reference for', sorted(k for k in out if k != 'input'))


def gen_vdif_corrupt():
    """File-surgery cases of the reference's own corrupt-file tests
    (vdif/tests/test_corrupt_files.py:13-156): a tripled sample.vdif with
    whole frames or byte ranges removed, read by the reference with
    verify='fix'.  Stored: the intact base file once, and per case the removed
    byte range(s), the sha256 of what the reference returns and which frames
    came back as fill."""
    import warnings
    with vdif.open(SAMPLE_VDIF, 'rs') as fs:
        bio = KeepBytesIO()
        with vdif.open(bio, 'ws', header0=fs.header0, nthread=8) as fw:
            data = fs.read()
            for i in range(3):
                fw.write(data)
    base = bio.value()
    full = np.concatenate([data, data, data]).reshape(-1, 8, 1)
    write_synth('vdif_triple', base, full, nthread=8, nchan=1, bps=2,
                complex_data=False, edv=3, samples_per_frame=20000, nframes=6,
                frame_rate=1600)
    fn = 5032
    cases = []
    frames = [36, (46, 48), [30, 45], (8, 16), 0, (4, 12)]
    for m in frames:
        if isinstance(m, tuple):
            idx = list(range(*m))
        elif isinstance(m, list):
            idx = m
        else:
            idx = [m]
        cases.append(dict(kind='frames', frames=idx,
                          remove=[[i * fn, (i + 1) * fn] for i in idx]))
    for lo, hi in ((fn * 26, fn * 26 + 1), (fn * 26 + 50, fn * 26 + 60),
                   (fn * 27 + 50, fn * 29 + 700), (fn * 31 + 10, fn * 31 + 20),
                   (fn * 32, fn * 32 + 10), (fn * 48 - 1, fn * 48)):
        cases.append(dict(kind='bytes', remove=[[lo, hi]]))
    # headers damaged in place (no bytes lost): sync pattern / frame length
    cases.append(dict(kind='overwrite', remove=[], flip=[fn * 27 + 21]))
    cases.append(dict(kind='overwrite', remove=[], flip=[fn * 12 + 22]))
    cases.append(dict(kind='overwrite', remove=[], flip=[fn * 29 + 20, fn * 30 + 23]))
    arr = np.frombuffer(base, np.uint8)
    for c in cases:
        keep = np.ones(len(arr), bool)
        for lo, hi in c['remove']:
            keep[lo:hi] = False
        work = arr.copy()
        for pos in c.get('flip', []):
            work[pos] ^= 0x55
        blob = work[keep].tobytes()
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                with vdif.open(io.BytesIO(blob), 'rs', squeeze=False) as fr:
                    got = fr.read()
        except Exception as exc:
            print('reference cannot read case', c, '->', type(exc).__name__)
            c['reference_fails'] = type(exc).__name__
            continue
        c['shape'] = list(got.shape)
        c['sha256'] = sha(got)
        # which (frame set, thread) blocks came back as zeros
        byframe = got.reshape(-1, 20000, 8).transpose(0, 2, 1).reshape(-1, 20000)
        ref = full[:got.shape[0]].reshape(-1, 20000, 8).transpose(0, 2, 1).reshape(-1, 20000)
        bad = [int(i) for i in range(len(byframe)) if not np.array_equal(byframe[i], ref[i])]
        assert all(np.all(byframe[i] == 0) for i in bad)
        c['zeroed'] = bad           # index = frame set * 8 + thread
    cases = [c for c in cases if 'reference_fails' not in c]
    with open(os.path.join(GOLD, 'vdif_corrupt_cases.json'), 'w') as f:
        json.dump(cases, f, indent=1)
    print('vdif corrupt:', len(cases), 'cases;', [c['zeroed'] for c in cases])


def gen_sequence():
    """Multi-file sequences through the reference (helpers/sequentialfile.py,
    baseband/tests/test_sequential_baseband.py): template names, byte-level
    reads across file boundaries, stream reads of a split file, and the files
    a sequence writer produces.  Stored as names / sizes / hashes."""
    from baseband.helpers import sequentialfile as sf
    out = {}
    with vdif.open(SAMPLE_VDIF, 'rb') as fh:
        vh = vdif.VDIFHeader.fromfile(fh)
    with open(SAMPLE_DADA, 'rb') as fh:
        dh = dada.DADAHeader.fromfile(fh)
    with open(SAMPLE_PUPPI, 'rb') as fh:
        gh = guppi.GUPPIHeader.fromfile(fh)
    out['names'] = {
        'plain': [sf.FileNameSequencer('a{file_nr:03d}.vdif')[10], 'a{file_nr:03d}.vdif'],
        'vdif_header': [sf.FileNameSequencer('obs.edv{edv:d}.{file_nr:05d}.vdif', vh)[10],
                        'obs.edv{edv:d}.{file_nr:05d}.vdif'],
        'dada': [[dada.DADAFileNameSequencer('{utc_start}.{obs_offset:016d}.000000.dada', dh)[i]
                  for i in (0, 1, 10)], '{utc_start}.{obs_offset:016d}.000000.dada'],
        'dada_date': [dada.DADAFileNameSequencer('{date}_{file_nr:03d}.dada',
                                                 {'DATE': "2018-01-01"})[10],
                      '{date}_{file_nr:03d}.dada'],
        'guppi': [guppi.GUPPIFileNameSequencer(
            'puppi_{stt_imjd}_{src_name}_{scannum}.{file_nr:04d}.raw', gh)[3],
            'puppi_{stt_imjd}_{src_name}_{scannum}.{file_nr:04d}.raw'],
    }
    tmp = tempfile.mkdtemp()
    # byte-level reads over three files of unequal size
    blob = open(SAMPLE_VDIF, 'rb').read()
    cuts = [0, 10000, 40001, len(blob)]
    names = []
    for i in range(3):
        names.append(os.path.join(tmp, 'part%d.vdif' % i))
        with open(names[-1], 'wb') as f:
            f.write(blob[cuts[i]:cuts[i + 1]])
    reads = []
    with sf.open(names, 'rb') as fh:
        size = fh.size
        for off, cnt in ((0, 100), (9990, 20), (9990, 40000), (40001, 10), (80000, 1000), (5, None)):
            fh.seek(off)
            d = fh.read(cnt) if cnt is not None else fh.read()
            reads.append(dict(offset=off, count=cnt, nbytes=len(d), tell=fh.tell(),
                              sha256=hashlib.sha256(d).hexdigest()))
    out['byte_reads'] = dict(cuts=cuts, size=size, reads=reads)
    with vdif.open(names, 'rs') as fs, vdif.open(SAMPLE_VDIF, 'rs') as f1:
        a = fs.read()
        assert np.array_equal(a, f1.read())
        fs.seek(19990)
        part = fs.read(30)
        out['vdif_split_stream'] = dict(shape=list(a.shape), sha256=sha(a),
                                        seek=19990, count=30, part_sha256=sha(part))
    # sequence writer: sample.vdif data into files of two frame sets each (x2 the data)
    with vdif.open(SAMPLE_VDIF, 'rs') as f1:
        data = f1.read()
        header0 = f1.header0
    template = os.path.join(tmp, 'w{file_nr:02d}.vdif')
    with vdif.open(template, 'ws', header0=header0, nthread=8, file_size=8 * 5032) as fw:
        fw.write(data)
        fw.write(data)
    written = sorted(f for f in os.listdir(tmp) if f.startswith('w'))
    out['vdif_sequence_write'] = dict(
        file_size=8 * 5032, files=written,
        sizes=[os.path.getsize(os.path.join(tmp, f)) for f in written],
        sha256=[hashlib.sha256(open(os.path.join(tmp, f), 'rb').read()).hexdigest()
                for f in written])
    with vdif.open(template, 'rs') as fr:
        back = fr.read()
    assert np.array_equal(back, np.concatenate([data, data]))
    # DADA: one frame per file (template with obs_offset), read back as a stream
    with dada.open(SAMPLE_DADA, 'rs') as f1:
        ddata = f1.read()
        dheader = f1.header0
    dtemplate = os.path.join(tmp, '{utc_start}.{obs_offset:016d}.000000.dada')
    with dada.open(dtemplate, 'ws', header0=dheader) as fw:
        fw.write(ddata)
        fw.write(ddata)
    dfiles = sorted(f for f in os.listdir(tmp) if f.endswith('.dada'))
    with dada.open(dtemplate, 'rs', UTC_START=dheader['UTC_START'],
                   OBS_OFFSET=dheader['OBS_OFFSET'], FILE_SIZE=dheader['FILE_SIZE']) as fr:
        dback = fr.read()
    out['dada_sequence'] = dict(files=dfiles,
                                sizes=[os.path.getsize(os.path.join(tmp, f)) for f in dfiles],
                                sha256=[hashlib.sha256(open(os.path.join(tmp, f), 'rb').read()).hexdigest()
                                        for f in dfiles],
                                shape=list(dback.shape), data_sha256=sha(dback))
    assert np.array_equal(dback, np.concatenate([ddata, ddata]))
    shutil.rmtree(tmp)
    with open(os.path.join(GOLD, 'sequence_cases.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('sequence:', out['names'], out['vdif_sequence_write']['sizes'], out['dada_sequence']['files'])


def gen_fixed_corrupt():
    """File-surgery cases of the reference's Mark 5B and Mark 4 corrupt-file
    tests (mark5b/tests/test_corrupt_files.py, mark4/tests/test_corrupt_files.py)
    read by the reference with verify='fix'.  Stored: the intact base files
    (compressed) and per case the removed byte range / replacement, the shape
    and sha256 of what the reference returns, and which frames came back as
    fill."""
    import warnings
    arrays, cases = {}, {}

    def run(opener, blob, kwargs, spf):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            with opener(io.BytesIO(blob), 'rs', **kwargs) as fr:
                n = fr.shape[0]
                data = fr.read()
        return data

    def record(lst, kind, blob, base_data, opener, kwargs, spf, **info):
        c = dict(kind=kind, **info)
        try:
            got = run(opener, blob, kwargs, spf)
        except Exception as exc:
            c['error'] = type(exc).__name__
            c['message'] = str(exc)[:200]
            lst.append(c)
            return
        c['shape'] = list(got.shape)
        c['sha256'] = sha(got)
        byframe = got.reshape(-1, spf * got.shape[-1])
        c['zeroed'] = [int(i) for i in range(len(byframe)) if not byframe[i].any()]
        lst.append(c)

    # ---- Mark 5B, sample file + 4 invalid frames (test_bad_bytes)
    sample = open(SAMPLE_MARK5B, 'rb').read()
    kw = dict(sample_rate=32 * u.MHz, kday=56000, nchan=8, bps=2)
    with mark5b.open(SAMPLE_MARK5B, 'rs', **kw) as fs:
        frame_rate = fs._frame_rate
        start_time = fs.start_time
        fs.read()
        frame3 = fs._frame
    tail = io.BytesIO()
    for i in range(4, 8):
        header = frame3.header.copy()
        header.set_time(start_time + i / frame_rate, frame_rate=frame_rate)
        header.update()
        frame3.__class__(header, frame3.payload, valid=False).tofile(tail)
    arrays['m5b_sample_tail'] = np.frombuffer(tail.getvalue(), np.uint8)
    lst = cases['m5b_sample'] = []
    kwj = dict(sample_rate=32e6, kday=56000, nchan=8, bps=2)
    for (lo, hi), rep in (((20032, 20033), b''), ((20096, 20100), b''), ((12000, 22000), b''),
                          ((30060, 30070), b''), ((40063, 40064), b''),
                          ((20032, 20033), b'\xff'), ((20032, 20036), b'\xff'),
                          ((20040, 20041), b'\xff')):
        blob = sample[:lo] + rep + sample[hi:] + tail.getvalue()
        record(lst, 'bytes', blob, None, mark5b.open, kw, 5000, remove=[lo, hi], replace=rep.hex())
    # ---- Mark 5B fake file (TestCorruptFile)
    time = Time('2010-11-12T13:14:15')
    header0 = mark5b.Mark5BHeader.fromvalues(time=time)
    data = np.repeat([[-1, 1], [-3, 3]], 10000, axis=0)
    bio = KeepBytesIO()
    with mark5b.open(bio, 'ws', header0=header0, sample_rate=100 * u.kHz, nchan=2) as fw:
        for _ in range(16):
            fw.write(data)
    fake = bio.value()
    arrays['m5b_fake'] = np.frombuffer(fake, np.uint8)
    kw = dict(nchan=2, sample_rate=100 * u.kHz, ref_time=time)
    lst = cases['m5b_fake'] = []
    fn = 10016
    todo = [((f0 * fn, f1 * fn), 'frames') for f0, f1 in ((1, 2), (3, 4), (5, 6), (7, 10))]
    todo += [((0, 8), 'start'), ((0, 9000), 'start'), ((0, 10012), 'start'), ((8, 10016), 'start')]
    todo += [((15 * fn + a, 15 * fn + b), 'end') for a, b in
             ((0, 10016), (0, 16), (8, 16), (0, 1), (10, 11), (15, 16), (20, 21), (10015, 10016))]
    todo += [((10016, 20032), 'middle'), ((20000, 20501), 'middle'), ((20032, 20048), 'middle')]
    for (lo, hi), kind in todo:
        record(lst, kind, fake[:lo] + fake[hi:], None, mark5b.open, kw, 20000, remove=[lo, hi], replace='')
    # ---- Mark 4 fake file
    header0 = mark4.Mark4Header.fromvalues(time=time, ntrack=16, nchan=2, fanout=4)
    data = np.zeros((2 * header0.frame_nbytes, 2))
    data.reshape(-1, 4, 2)[160:] = [[-1, 1], [-3, 3], [1, -1], [3, -3]]
    bio = KeepBytesIO()
    with mark4.open(bio, 'ws', header0=header0, sample_rate=100 * u.kHz) as fw:
        for _ in range(8):
            fw.write(data)
    fake = bio.value()
    arrays['m4_fake'] = np.frombuffer(fake, np.uint8)
    kw = dict(sample_rate=100 * u.kHz, ref_time=time)
    lst = cases['m4_fake'] = []
    fn = 40000
    todo = [((f0 * fn, f1 * fn), 'frames') for f0, f1 in ((1, 2), (3, 4), (3, 5))]
    todo += [((7 * fn + a, 7 * fn + b), 'end') for a, b in
             ((0, 40000), (0, 320), (8, 16), (0, 1), (10, 11), (319, 320), (400, 401), (39999, 40000))]
    todo += [((40000, 80000), 'middle'), ((78000, 82000), 'middle'), ((80010, 80100), 'middle')]
    for (lo, hi), kind in todo:
        record(lst, kind, fake[:lo] + fake[hi:], None, mark4.open, kw, 80000, remove=[lo, hi], replace='')
    # duplicated data: 'excess data' error in the reference
    record(lst, 'duplicate', fake[:100000] + fake[40000:], None, mark4.open, kw, 80000,
           remove=[100000, 40000], replace='')
    np.savez_compressed(os.path.join(GOLD, 'fixed_corrupt_files.npz'), **arrays)
    with open(os.path.join(GOLD, 'fixed_corrupt_cases.json'), 'w') as f:
        json.dump(cases, f, indent=1)
    for k, lst in cases.items():
        print(k, [(c['kind'], c.get('shape', c.get('error')), c.get('zeroed')) for c in lst])


def gen_info():
    """``info`` of the reference for every sample file, as plain JSON values
    (base/file_info.py; */file_info.py)."""
    def plain(v):
        if isinstance(v, Time):
            return v.isot
        if isinstance(v, u.Quantity):
            return float(v.to_value(u.Hz))
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (tuple, list)):
            return [plain(x) for x in v]
        if isinstance(v, (np.integer, np.bool_)):
            return v.item()
        if isinstance(v, Exception):
            return repr(v)
        return v

    out = {}
    todo = [('sample_vdif', vdif, SAMPLE_VDIF, {}), ('sample_mwa_vdif', vdif, SAMPLE_MWA_VDIF, {}),
            ('sample_arochime_vdif', vdif, SAMPLE_AROCHIME_VDIF, {}),
            ('sample_bps1_vdif', vdif, SAMPLE_BPS1_VDIF, {}),
            ('sample_m5b_bare', mark5b, SAMPLE_MARK5B, {}),
            ('sample_m5b', mark5b, SAMPLE_MARK5B, dict(kday=56000, nchan=8)),
            ('sample_m4_bare', mark4, SAMPLE_MARK4, {}),
            ('sample_m4', mark4, SAMPLE_MARK4, dict(ntrack=64, decade=2010)),
            ('sample_m4_32', mark4, SAMPLE_MARK4_32TRACK, dict(ntrack=32, decade=2010)),
            ('sample_dada', dada, SAMPLE_DADA, {}), ('sample_puppi', guppi, SAMPLE_PUPPI, {})]
    for key, mod, path, kw in todo:
        with mod.open(path, 'rb', **kw) as fh:
            d = plain(fh.info())
        d.pop('file_info', None)
        entry = dict(file=os.path.basename(path), kwargs={k: v for k, v in kw.items()}, file_info=d)
        if not d.get('missing'):
            skw = dict(kw)
            if mod is mark5b:
                skw['sample_rate'] = 32 * u.MHz
            if mod is mark4:
                skw['sample_rate'] = 32 * u.MHz
            if 'mwa' in key:
                skw['sample_rate'] = 1.28 * u.MHz
            if 'arochime' in key:
                skw['sample_rate'] = 800. / 1024. / 2. * u.MHz
            if 'bps1' in key:
                skw['sample_rate'] = 8 * u.MHz
            try:
                with mod.open(path, 'rs', **skw) as fs:
                    sd = plain(fs.info())
            except Exception as exc:
                print('no stream info for', key, repr(exc)[:80])
                out[key] = entry
                continue
            sd.pop('file_info', None)
            entry['stream_info'] = sd
        out[key] = entry
    with open(os.path.join(GOLD, 'info_cases.json'), 'w') as f:
        json.dump(out, f, indent=1, default=str)
    for k, v in out.items():
        print(k, v['file_info'])


def gen_gsb_writer():
    """GSB stream writer of the reference (gsb/base.py:388-447) re-writing the
    two sample observations: the timestamp text and the raw bytes it produces."""
    out = {}
    tmp = tempfile.mkdtemp()
    with gsb.open(SAMPLE_GSB_RAWDUMP_HEADER, 'rs', raw=SAMPLE_GSB_RAWDUMP,
                  samples_per_frame=8192) as fr:
        data = fr.read()
        ts, raw = os.path.join(tmp, 'r.timestamp'), os.path.join(tmp, 'r.dat')
        with gsb.open(ts, 'ws', raw=raw, header0=fr.header0, sample_rate=fr.sample_rate,
                      samples_per_frame=8192) as fw:
            fw.write(data)
    out['rawdump_ts'] = np.frombuffer(open(ts, 'rb').read(), np.uint8)
    out['rawdump_raw'] = np.fromfile(raw, np.uint8)
    assert np.array_equal(out['rawdump_raw'], np.fromfile(SAMPLE_GSB_RAWDUMP, np.uint8))
    with gsb.open(SAMPLE_GSB_PHASED_HEADER, 'rs', raw=SAMPLE_GSB_PHASED,
                  samples_per_frame=8) as fr:
        data = fr.read()
        ts = os.path.join(tmp, 'p.timestamp')
        raws = [[os.path.join(tmp, 'p%d%d.dat' % (p, f)) for f in range(2)] for p in range(2)]
        with gsb.open(ts, 'ws', raw=raws, sample_rate=fr.sample_rate, samples_per_frame=8,
                      **fr.header0) as fw:
            fw.write(data)
    out['phased_ts'] = np.frombuffer(open(ts, 'rb').read(), np.uint8)
    for p in range(2):
        for f in range(2):
            out['phased_raw%d%d' % (p, f)] = np.fromfile(raws[p][f], np.uint8)
            assert np.array_equal(out['phased_raw%d%d' % (p, f)],
                                  np.fromfile(SAMPLE_GSB_PHASED[p][f], np.uint8))
    # non-integer samples, one pol / one file phased, 4-bit rawdump clipping
    t0 = Time('2015-06-01T01:02:03.251658240', precision=9)
    rng = np.random.default_rng(77)
    x = (rng.standard_normal(3 * 64) * 5).astype(np.float32)
    ts, raw = os.path.join(tmp, 'r2.timestamp'), os.path.join(tmp, 'r2.dat')
    with gsb.open(ts, 'ws', raw=raw, time=t0, samples_per_frame=64, sample_rate=1 * u.kHz) as fw:
        fw.write(x)
    out['raw2_in'] = x
    out['raw2_ts'] = np.frombuffer(open(ts, 'rb').read(), np.uint8)
    out['raw2_raw'] = np.fromfile(raw, np.uint8)
    z = (rng.standard_normal((40, 4)) * 60 + 1j * rng.standard_normal((40, 4)) * 60).astype(np.complex64)
    ts, raw = os.path.join(tmp, 'p2.timestamp'), os.path.join(tmp, 'p2.dat')
    with gsb.open(ts, 'ws', raw=raw, header_mode='phased', time=t0, seq_nr=9998, mem_block=6,
                  samples_per_frame=8, nchan=4, sample_rate=2 * u.kHz) as fw:
        fw.write(z)
    out['ph2_in'] = z
    out['ph2_ts'] = np.frombuffer(open(ts, 'rb').read(), np.uint8)
    out['ph2_raw'] = np.fromfile(raw, np.uint8)
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(GOLD, 'gsb_writer_cases.npz'), **out)
    print('gsb writer:', {k: v.shape for k, v in out.items()})
    print(out['rawdump_ts'].tobytes().decode()[:200])
    print(out['phased_ts'].tobytes().decode()[:300])
    print(out['ph2_ts'].tobytes().decode())
    print(out['raw2_ts'].tobytes().decode())


def gen_stream_fuzz():
    """Random stream-API calls through the reference: for every sample file a
    set of (squeeze, subset) openings and (seek, read) sequences; stored are
    shapes and digests of what comes back (base/base.py:706-717,876-969)."""
    rng = np.random.default_rng(20261002)
    files = [
        ('vdif', 'samples/sample.vdif', SAMPLE_VDIF, {}, (8, 1)),
        ('vdif', 'samples/sample_arochime.vdif', SAMPLE_AROCHIME_VDIF,
         dict(sample_rate=800. / 1024. / 2. * u.MHz), (2, 1024)),
        ('vdif', 'samples/sample_mwa.vdif', SAMPLE_MWA_VDIF, dict(sample_rate=1.28 * u.MHz), (1, 2)),
        ('mark5b', 'samples/sample.m5b', SAMPLE_MARK5B,
         dict(kday=56000, nchan=8, sample_rate=32 * u.MHz), (8,)),
        ('mark4', 'samples/sample.m4', SAMPLE_MARK4, dict(ntrack=64, decade=2010, sample_rate=32 * u.MHz), (8,)),
        ('mark4', 'samples/sample_32track_fanout2.m4', SAMPLE_MARK4_32TRACK_FANOUT2,
         dict(ntrack=32, decade=2010), (8,)),
        ('dada', 'samples/sample.dada', SAMPLE_DADA, {}, (2, 1)),
        ('guppi', 'samples/sample_puppi.raw', SAMPLE_PUPPI, {}, (2, 4)),
        ('gsb', 'samples/gsb/sample_gsb_rawdump.timestamp', SAMPLE_GSB_RAWDUMP_HEADER,
         dict(raw=SAMPLE_GSB_RAWDUMP, samples_per_frame=8192), (1,)),
        ('gsb', 'samples/gsb/sample_gsb_phased.timestamp', SAMPLE_GSB_PHASED_HEADER,
         dict(raw=SAMPLE_GSB_PHASED, samples_per_frame=8), (2, 512)),
    ]
    mods = dict(vdif=vdif, mark5b=mark5b, mark4=mark4, dada=dada, guppi=guppi, gsb=gsb)

    def pick(n):
        kind = rng.integers(0, 4)
        if n == 1 or kind == 0:
            return None                                  # no selection on this axis
        if kind == 1:
            return int(rng.integers(0, n))
        if kind == 2:
            a = int(rng.integers(0, n - 1))
            b = int(rng.integers(a + 1, n + 1))
            return ['slice', a, b, int(rng.integers(1, 3))]
        k = int(rng.integers(1, min(n, 4) + 1))
        return [int(v) for v in rng.choice(n, size=k, replace=False)]

    def decode(x):
        return slice(x[1], x[2], x[3]) if isinstance(x, list) and x and x[0] == 'slice' else x

    out = []
    for fmt, rel, path, kw, shape in files:
        for trial in range(6):
            squeeze = bool(rng.integers(0, 2))
            eff = tuple(d for d in shape if d > 1) if squeeze else shape
            sub = []
            for n in eff:
                sub.append(pick(n))
            while sub and sub[-1] is None:
                sub.pop()
            if any(v is None for v in sub):              # only trailing axes may be left out
                sub = [v if v is not None else ['slice', 0, n, 1] for v, n in zip(sub, eff)]
            # two list selections are broadcast together by numpy; keep at most one
            seen_list = False
            for i, v in enumerate(sub):
                if isinstance(v, list) and v and v[0] != 'slice':
                    if seen_list:
                        sub[i] = int(v[0])
                    seen_list = True
            subset = tuple(decode(v) for v in sub)
            kwargs = dict(kw, squeeze=squeeze, subset=subset)
            try:
                with mods[fmt].open(path, 'rs', **kwargs) as fh:
                    n = fh.shape[0]
                    ops = []
                    for _ in range(4):
                        off = int(rng.integers(0, n))
                        cnt = int(rng.integers(1, min(n - off, 30000) + 1))
                        fh.seek(off)
                        d = fh.read(cnt)
                        ops.append(dict(seek=off, count=cnt, shape=list(d.shape), sha256=sha(d),
                                        tell=int(fh.tell())))
                    case = dict(fmt=fmt, file=rel, squeeze=squeeze, subset=sub,
                                shape=list(fh.shape), sample_shape=list(fh.sample_shape), ops=ops)
            except Exception as exc:
                case = dict(fmt=fmt, file=rel, squeeze=squeeze, subset=sub,
                            error=type(exc).__name__)
            out.append(case)
    with open(os.path.join(GOLD, 'stream_fuzz_cases.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('stream fuzz:', len(out), 'cases;', sum('error' in c for c in out), 'raise in the reference')
    for c in out[:8]:
        print(c['fmt'], c['squeeze'], c['subset'], c.get('shape'), c.get('error'))


def gen_item_fuzz():
    """Random item access on payloads / frames / frame sets of the sample files
    through the reference (base/payload.py:226-330, base/frame.py:191-199,
    vdif/frame.py:402-434, mark4/frame.py:152-263, guppi/payload.py:104-133)."""
    rng = np.random.default_rng(424242)

    def rand_index(n):
        kind = rng.integers(0, 5)
        if kind == 0:
            return int(rng.integers(-n, n))
        if kind == 1:
            return ['slice', None, None, None]
        a = int(rng.integers(0, n))
        b = int(rng.integers(a, n + 1))
        step = None if kind < 4 else int(rng.integers(2, 6))
        return ['slice', a, b, step]

    def rand_sub(n):
        kind = rng.integers(0, 3)
        if kind == 0:
            return int(rng.integers(0, n))
        a = int(rng.integers(0, n))
        b = int(rng.integers(a + 1, n + 1))
        return ['slice', a, b, None]

    def dec(x):
        return slice(x[1], x[2], x[3]) if isinstance(x, list) else x

    objs = []
    with vdif.open(SAMPLE_VDIF, 'rb') as fh:
        fr = fh.read_frame()
        objs.append(('vdif_frame', 'samples/sample.vdif', fr))
        objs.append(('vdif_payload', 'samples/sample.vdif', fr.payload))
        fh.seek(0)
        objs.append(('vdif_frameset', 'samples/sample.vdif', fh.read_frameset()))
    with vdif.open(SAMPLE_AROCHIME_VDIF, 'rb') as fh:
        objs.append(('vdif_frame', 'samples/sample_arochime.vdif', fh.read_frame()))
    with vdif.open(SAMPLE_MWA_VDIF, 'rb') as fh:
        objs.append(('vdif_payload', 'samples/sample_mwa.vdif', fh.read_frame().payload))
    with mark5b.open(SAMPLE_MARK5B, 'rb', kday=56000, nchan=8) as fh:
        objs.append(('mark5b_frame', 'samples/sample.m5b', fh.read_frame()))
    with mark4.open(SAMPLE_MARK4, 'rb', ntrack=64, decade=2010) as fh:
        fh.find_header()
        objs.append(('mark4_frame', 'samples/sample.m4', fh.read_frame()))
    with dada.open(SAMPLE_DADA, 'rb') as fh:
        objs.append(('dada_frame', 'samples/sample.dada', fh.read_frame()))
    with guppi.open(SAMPLE_PUPPI, 'rb') as fh:
        fr = fh.read_frame(memmap=False)
        objs.append(('guppi_frame', 'samples/sample_puppi.raw', fr))
        objs.append(('guppi_payload', 'samples/sample_puppi.raw', fr.payload))
    out = []
    for kind, rel, obj in objs:
        shape = obj.shape
        items = []
        for _ in range(12):
            item = [rand_index(shape[0])]
            for n in shape[1:]:
                if rng.integers(0, 2):
                    item.append(rand_sub(n))
                else:
                    break
            key = tuple(dec(v) for v in item)
            key = key[0] if len(key) == 1 and rng.integers(0, 2) else key
            try:
                d = np.asarray(obj[key])
                items.append(dict(item=item, bare=not isinstance(key, tuple), shape=list(d.shape),
                                  dtype=str(d.dtype), sha256=sha(np.ascontiguousarray(d))))
            except Exception as exc:
                items.append(dict(item=item, bare=not isinstance(key, tuple), error=type(exc).__name__))
        out.append(dict(kind=kind, file=rel, shape=list(shape), items=items))
    with open(os.path.join(GOLD, 'item_fuzz_cases.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('item fuzz:', [(c['kind'], c['shape'], sum('error' in i for i in c['items'])) for c in out])


def gen_locate():
    """locate_frames / find_header of the reference (base/base.py:181-368,
    vdif/base.py:216-316, mark5b/base.py:126-155, mark4/base.py:110-166) at the
    positions its own tests probe plus random ones; stored: the calls and the
    returned locations / header offsets."""
    rng = np.random.default_rng(99)
    out = {}

    def probe(fh, n, header0, calls, kwlist):
        for kw in kwlist:
            for pos in [0, 10, 16, n - 20, n] + [int(v) for v in rng.integers(0, n, 12)]:
                fh.seek(pos)
                args = (header0,) if header0 is not None else ()
                try:
                    loc = fh.locate_frames(*args, **kw)
                except Exception as exc:
                    loc = type(exc).__name__
                fh.seek(pos)
                try:
                    fh.find_header(*args, **{k: v for k, v in kw.items()})
                    found = fh.tell()
                except Exception as exc:
                    found = type(exc).__name__
                calls.append(dict(pos=pos, kwargs={k: (list(v) if isinstance(v, tuple) else v)
                                                   for k, v in kw.items()},
                                  locations=loc, found=found))

    kws = [dict(), dict(forward=False), dict(check=(-1, 1)), dict(forward=False, check=(-1, 1)),
           dict(maximum=20000), dict(forward=False, maximum=30000, check=(1, 2)), dict(check=None),
           dict(maximum=0)]
    with vdif.open(SAMPLE_VDIF, 'rb') as fh:
        header0 = vdif.VDIFHeader.fromfile(fh)
        n = fh.seek(0, 2)
        calls = []
        probe(fh, n, header0, calls, kws)
        # explicit pattern / mask / offset forms (vdif/tests/test_vdif.py:694-727)
        extra = []
        fh.seek(0)
        extra.append(dict(pos=0, form='sync', locations=fh.locate_frames(pattern=header0['sync_pattern'], offset=20)))
        fh.seek(0, 2)
        extra.append(dict(pos=n, form='sync_back',
                          locations=fh.locate_frames(pattern=header0['sync_pattern'], offset=20, forward=False)))
        fh.seek(10)
        mask = [0, 0, 0xffffffff, 0xfc00ffff, 0xffffffff, 0, 0, 0]
        extra.append(dict(pos=10, form='words_mask',
                          locations=fh.locate_frames(pattern=header0.words, mask=mask, frame_nbytes=5032)))
        out['vdif'] = dict(file='samples/sample.vdif', calls=calls, extra=extra)
    blob = open(SAMPLE_VDIF, 'rb').read()
    gap = blob[:5100] + blob[10000:]
    with vdif.open(io.BytesIO(gap), 'rb') as fh:
        calls = []
        probe(fh, len(gap), header0, calls, kws[:4])
        out['vdif_gap'] = dict(file='samples/sample.vdif', cut=[5100, 10000], calls=calls)
    with mark5b.open(SAMPLE_MARK5B, 'rb', kday=56000, nchan=8) as fh:
        n = fh.seek(0, 2)
        calls = []
        probe(fh, n, None, calls, kws)
        out['mark5b'] = dict(file='samples/sample.m5b', calls=calls)
    with mark4.open(SAMPLE_MARK4, 'rb', ntrack=64, decade=2010) as fh:
        n = fh.seek(0, 2)
        calls = []
        probe(fh, n, None, calls, [dict(), dict(forward=False), dict(check=(-1, 1)), dict(maximum=400000),
                                   dict(forward=False, maximum=400000), dict(check=None)])
        out['mark4'] = dict(file='samples/sample.m4', calls=calls)
    with open(os.path.join(GOLD, 'locate_cases.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('locate:', {k: len(v['calls']) for k, v in out.items()})
    print(out['vdif']['extra'])


def gen_header_fuzz():
    """Headers built from keywords by the reference (vdif/header.py:188-290,
    mark5b/header.py:120-176, mark4/header.py:456-538): words and derived
    properties for seeded random keyword sets."""
    rng = np.random.default_rng(5150)
    out = dict(vdif=[], mark5b=[], mark4=[])
    t_lo = Time('2001-01-01T00:00:00').unix
    for i in range(60):
        edv = [False, 0, 1, 2, 3, 0xab][i % 6]
        bps = int(rng.choice([1, 2, 4, 8])) if edv != 0xab else int(rng.choice([1, 2]))
        cplx = bool(rng.integers(0, 2)) if edv not in (0xab,) else False
        nchan = int(2 ** rng.integers(0, 6))
        kw = dict(bps=bps, complex_data=cplx, nchan=nchan, thread_id=int(rng.integers(0, 1024)),
                  station=(int(rng.integers(0, 65536)) if rng.integers(0, 2) else
                           ''.join(chr(int(c)) for c in rng.integers(65, 91, 2))),
                  invalid_data=bool(rng.integers(0, 2)))
        if edv == 3:
            pass                                   # fixed frame length
        elif edv == 0xab:
            kw['frame_length'] = 1254
        else:
            kw['payload_nbytes'] = int(rng.integers(1, 1000)) * 8
        frame_rate = int(rng.choice([100, 1000, 1600, 6400, 25600]))
        tsec = float(rng.integers(int(t_lo), int(t_lo) + 20 * 365 * 86400))
        nfr = int(rng.integers(0, frame_rate))
        time = Time(tsec, format='unix', precision=9) + nfr / frame_rate * u.s
        kw['time'] = time
        if edv in (1, 3):
            h0 = vdif.VDIFHeader.fromvalues(edv=edv, **{k: v for k, v in kw.items() if k != 'time'})
            kw['sample_rate'] = frame_rate * h0.samples_per_frame * u.Hz
        else:
            kw['frame_rate'] = frame_rate * u.Hz
        if edv == 2:
            kw.pop('complex_data')
        try:
            h = vdif.VDIFHeader.fromvalues(edv=edv, **kw)
        except Exception as exc:
            continue
        rec = {k: (v if not isinstance(v, (Time, u.Quantity)) else None) for k, v in kw.items()}
        rec.pop('time')
        rec.pop('sample_rate', None)
        rec.pop('frame_rate', None)
        out['vdif'].append(dict(
            edv=(-1 if edv is False else edv), kwargs=rec, time_unix_ns=int(round(tsec)) * 10**9 + round(nfr * 10**9 / frame_rate),
            frame_rate=frame_rate, words=[int(w) for w in h.words], nbytes=int(h.nbytes),
            frame_nbytes=int(h.frame_nbytes), payload_nbytes=int(h.payload_nbytes), bps=int(h.bps),
            nchan=int(h.nchan), samples_per_frame=int(h.samples_per_frame),
            complex_data=bool(h.complex_data), station=h.station if isinstance(h.station, str) else int(h.station),
            sample_rate=(float(h.sample_rate.to_value(u.Hz)) if edv in (1, 3) else None)))
    for i in range(25):
        frame_rate = int(rng.choice([400, 1600, 6400, 25600]))
        tsec = float(rng.integers(int(t_lo), int(t_lo) + 20 * 365 * 86400))
        nfr = int(rng.integers(0, frame_rate))
        time = Time(tsec, format='unix', precision=9) + nfr / frame_rate * u.s
        user = int(rng.integers(0, 65536))
        h = mark5b.Mark5BHeader.fromvalues(time=time, frame_rate=frame_rate * u.Hz, user=user,
                                           internal_tvg=bool(i % 2))
        out['mark5b'].append(dict(time_unix_ns=int(round(tsec)) * 10**9 + round(nfr * 10**9 / frame_rate),
                                  frame_rate=frame_rate, user=user, internal_tvg=bool(i % 2),
                                  words=[int(w) for w in h.words], kday=int(h.kday), jday=int(h.jday),
                                  seconds=int(h.seconds), frame_nr=int(h['frame_nr'])))
    for i in range(25):
        ntrack, fanout = [(64, 4), (64, 2), (32, 4), (32, 2), (16, 4)][i % 5]
        tsec = float(rng.integers(int(Time('2010-01-01').unix), int(Time('2019-12-30').unix)))
        ms = int(rng.integers(0, 800)) * 1.25
        time = Time(tsec, format='unix', precision=9) + ms * u.ms
        try:
            h = mark4.Mark4Header.fromvalues(ntrack=ntrack, time=time, bps=2, fanout=fanout, nsb=1)
        except Exception as exc:
            continue
        out['mark4'].append(dict(ntrack=ntrack, fanout=fanout, time_unix_ns=int(round(tsec)) * 10**9 + int(round(ms * 1e6)),
                                 words=np.asarray(h.words).astype(np.uint64).tolist(), nchan=int(h.nchan),
                                 samples_per_frame=int(h.samples_per_frame), time_isot=h.time.isot))
    # Mark 5B frames wrapped as VDIF EDV 0xab (vdif/frame.py:104-128), and update()
    out['mark5b_to_vdif'] = []
    with mark5b.open(SAMPLE_MARK5B, 'rb', kday=56000, nchan=8) as fh:
        for k in range(4):
            m5 = fh.read_frame()
            vf = vdif.VDIFFrame.from_mark5b_frame(m5)
            b = io.BytesIO()
            vf.tofile(b)
            out['mark5b_to_vdif'].append(dict(words=[int(w) for w in vf.header.words],
                                              frame_sha256=hashlib.sha256(b.getvalue()).hexdigest(),
                                              valid=bool(vf.valid)))
    h = vdif.VDIFHeader.fromvalues(edv=1, bps=2, nchan=4, complex_data=True, payload_nbytes=4000,
                                   station='Ab', time=Time('2015-06-07T08:09:10'), sample_rate=16 * u.MHz)
    h.update(thread_id=7, bps=4, nchan=2, time=Time('2015-06-07T08:09:11.25'), frame_rate=8000 * u.Hz)
    out['vdif_update'] = [int(w) for w in h.words]
    m = mark5b.Mark5BHeader.fromvalues(time=Time('2015-06-07T08:09:10'), user=5)
    m.update(user=77, time=Time('2015-06-07T08:09:10.5'), frame_rate=6400 * u.Hz)
    out['mark5b_update'] = [int(w) for w in m.words]
    with open(os.path.join(GOLD, 'header_fuzz_cases.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('header fuzz:', {k: len(v) for k, v in out.items()})


def gen_mark4_header():
    """Mark 4 header construction by the reference (mark4/header.py:345-373,
    456-533, 558-739): words for keyword combinations (bps/nchan,
    fanout/samples_per_frame, nsb, converters), update(), fromkeys() and the
    class / stream invariant patterns."""
    out = dict(fromvalues=[], update=[], patterns=[])
    time = Time('2014-06-15T12:34:56.0125', precision=9)
    tns = 1402835696 * 10**9 + 12500000
    assert abs(time.unix - tns / 1e9) < 1e-6
    combos = []
    for ntrack in (16, 32, 64):
        for fanout in (1, 2, 4):
            for bps in (1, 2):
                for nsb in (1, 2):
                    combos.append(dict(ntrack=ntrack, fanout=fanout, bps=bps, nsb=nsb))
    combos += [dict(ntrack=64, fanout=4, nchan=8), dict(ntrack=32, samples_per_frame=80000, bps=2),
               dict(ntrack=32, samples_per_frame=40000, nchan=16, system_id=108),
               dict(ntrack=64, fanout=4, bps=2, nsb=2, converters=[3, 1, 2, 0]),
               dict(ntrack=64, fanout=4, bps=2, nsb=1, converters=[7, 6, 5, 4, 3, 2, 1, 0]),
               dict(ntrack=32, fanout=2, bps=2,
                    converters=dict(converter=[0, 1, 2, 3, 4, 5, 6, 7],
                                    lsb=[True, False, True, False, False, True, False, True])),
               dict(ntrack=64, fanout=4, bps=2, nsb=1, converters=[1, 2, 3]),
               dict(ntrack=32, fanout=3, bps=2), dict(ntrack=32, fanout=4, bps=3),
               dict(ntrack=32, samples_per_frame=12345, bps=2),
               dict(ntrack=64, fanout=4, bps=2, track_id=np.arange(64).tolist()),
               dict(ntrack=64, fanout=4, bps=2, nosuchkey=1)]
    for kw in combos:
        rec = dict(kwargs=kw)
        kw = {k: (np.array(v) if isinstance(v, list) else v) for k, v in kw.items()}
        if isinstance(kw.get('converters'), dict):
            kw['converters'] = {k: np.array(v) for k, v in kw['converters'].items()}
        try:
            h = mark4.Mark4Header.fromvalues(time=time, **kw)
        except Exception as exc:
            rec['error'] = type(exc).__name__
        else:
            rec.update(words=np.asarray(h.words).astype(np.uint64).tolist(),
                       nchan=int(h.nchan), bps=int(h.bps), fanout=int(h.fanout), nsb=int(h.nsb),
                       samples_per_frame=int(h.samples_per_frame),
                       track_id=np.asarray(h.track_id).tolist(),
                       converter=h.converters['converter'].tolist(),
                       lsb=h.converters['lsb'].tolist(), decade=int(h.decade),
                       fraction=float(np.asarray(h.fraction).ravel()[0]))
            k = mark4.Mark4Header.fromkeys(h.ntrack, h.decade, **h)
            assert k == h
        out['fromvalues'].append(rec)
    h = mark4.Mark4Header.fromvalues(ntrack=32, time=time, bps=2, fanout=4)
    for kw in (dict(system_id=12), dict(bps=1), dict(nsb=2), dict(fanout=2, bps=2, nsb=1),
               dict(time_ns=tns + 3 * 10**9 + 250 * 10**6), dict(crc=0x123, verify=False),
               dict(fraction=0.99875), dict(nchan=16), dict(bcd_day=0x123, nsb=2, fanout=1)):
        m = h.copy()
        kk = dict(kw)
        if 'time_ns' in kk:
            kk['time'] = time + (kk.pop('time_ns') - tns) / 1e9 * u.s
        try:
            m.update(**kk)
        except Exception as exc:
            out['update'].append(dict(kwargs=kw, error=type(exc).__name__))
        else:
            out['update'].append(dict(kwargs=kw, words=np.asarray(m.words).astype(np.uint64).tolist()))
    for ntrack in (16, 32, 64):
        pat, mask = mark4.Mark4Header.invariant_pattern(ntrack=ntrack)
        hh = mark4.Mark4Header.fromvalues(ntrack=ntrack, time=time, bps=2, fanout=4, system_id=108)
        ipat, imask = hh.invariant_pattern()
        out['patterns'].append(dict(ntrack=ntrack, pattern=np.asarray(pat).astype(np.uint64).tolist(),
                                    mask=np.asarray(mask).astype(np.uint64).tolist(),
                                    stream_pattern=np.asarray(ipat).astype(np.uint64).tolist(),
                                    stream_mask=np.asarray(imask).astype(np.uint64).tolist()))
    out['time_unix_ns'] = tns
    with open(os.path.join(GOLD, 'mark4_header_cases.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('mark4 header:', {k: (len(v) if isinstance(v, list) else v) for k, v in out.items()},
          'errors', sum('error' in r for r in out['fromvalues']))


def gen_setitem():
    """``obj[item] = values`` on payloads, frames and frame sets through the
    reference (base/payload.py:332-347, base/frame.py:203-207,
    mark4/frame.py:265-295, vdif/frame.py:436-486, guppi/payload.py:112-140):
    initial words, the assignments (items + values) and the words afterwards."""
    rng = np.random.default_rng(20240611)
    arrays, meta = {}, []

    def dec(x):
        return slice(x[1], x[2], x[3]) if isinstance(x, list) else x

    def values(shape, cplx, scale):
        v = rng.normal(0., scale, size=shape)
        if cplx:
            v = v + 1j * rng.normal(0., scale, size=shape)
            return v.astype('c8')
        return v.astype('f4')

    def run(name, obj, words_of, ops, cplx, scale, recipe):
        arrays[name + '_words0'] = np.concatenate([np.asarray(w).view(np.uint8).ravel() for w in words_of(obj)])
        recs = []
        for j, item in enumerate(ops):
            key = tuple(dec(v) for v in item)
            key = key[0] if len(key) == 1 else key
            shape = np.empty(obj.shape, bool)[key].shape
            v = values(shape, cplx, scale)
            arrays['{}_v{}'.format(name, j)] = v
            try:
                obj[key] = v
                recs.append(dict(item=item))
            except Exception as exc:
                recs.append(dict(item=item, error=type(exc).__name__))
        arrays[name + '_words1'] = np.concatenate([np.asarray(w).view(np.uint8).ravel() for w in words_of(obj)])
        meta.append(dict(name=name, recipe=recipe, shape=list(obj.shape), ops=recs))

    S = lambda a, b, c=None: ['slice', a, b, c]
    # VDIF payloads
    for name, bps, cplx, nchan, n in (('vdif_p2r4', 2, False, 4, 64), ('vdif_p4c2', 4, True, 2, 48),
                                      ('vdif_p1r1', 1, False, 1, 256), ('vdif_p8r16', 8, False, 16, 20)):
        d = values((n, nchan), cplx, 2. if bps < 8 else 30.)
        pl = vdif.VDIFPayload.fromdata(d, bps=bps)
        pl.words = pl.words.copy()
        run(name, pl, lambda o: [o.words],
            [[3], [S(5, 17)], [S(2, 40, 3), min(1, nchan - 1)], [S(None, None)], [-1, S(0, 1)], [S(7, 8)]],
            cplx, 2. if bps < 8 else 30., dict(kind='vdif_payload', bps=bps, complex_data=cplx, nchan=nchan))
    # Mark 5B payload (fixed 10000 bytes): 8 channels, 2 bit -> 5000 samples
    d = values((5000, 8), False, 2.)
    pl = mark5b.Mark5BPayload.fromdata(d, bps=2)
    pl.words = pl.words.copy()
    run('mark5b_p2', pl, lambda o: [o.words], [[11], [S(100, 228)], [S(3, 999, 7), S(2, 5)], [S(4990, None)]],
        False, 2., dict(kind='mark5b_payload', bps=2, nchan=8))
    # Mark 4 frame: assignments reaching into the part under the header
    time = Time('2014-06-15T12:34:56.0125', precision=9)
    h4 = mark4.Mark4Header.fromvalues(ntrack=32, time=time, bps=2, fanout=4)
    d = values((h4.samples_per_frame, h4.nchan), False, 2.)
    fr = mark4.Mark4Frame.fromdata(d, h4)
    fr.payload.words = fr.payload.words.copy()
    run('mark4_f32', fr, lambda o: [o.payload.words],
        [[S(0, 700)], [650, 1], [S(10000, 10100, 7)], [S(630, 650), S(1, 3)], [S(100, 200)], [79999]],
        False, 2., dict(kind='mark4_frame', ntrack=32, fanout=4, bps=2,
                        time_unix_ns=1402835696 * 10**9 + 12500000))
    # DADA payloads (standard and MKBF heaps)
    d = values((64, 2, 4), True, 30.)
    hd = dada.DADAHeader.fromvalues(time=Time('2013-07-02T01:39:20'), samples_per_frame=64,
                                    sample_rate=16 * u.MHz, bps=8, complex_data=True, sample_shape=(2, 4))
    pl = dada.DADAPayload.fromdata(d, header=hd)
    pl.words = pl.words.copy()
    run('dada_c', pl, lambda o: [o.words], [[3], [S(5, 17)], [S(2, 40, 3), 1], [S(8, 12), S(None, None), 2], [S(None, None)]],
        True, 30., dict(kind='dada_payload', sample_shape=[2, 4], complex_data=True))
    # GUPPI payloads: channels first and time first
    for name, cf in (('guppi_cf', True), ('guppi_tf', False)):
        d = values((96, 2, 4), True, 30.)
        pl = guppi.GUPPIPayload.fromdata(d, bps=8, channels_first=cf)
        pl.words = pl.words.copy()
        run(name, pl, lambda o: [o.words], [[3], [S(5, 17)], [S(2, 40, 3), 1], [S(8, 12), S(None, None), 2], [S(90, None)]],
            True, 30., dict(kind='guppi_payload', sample_shape=[2, 4], channels_first=cf))
    # GSB rawdump 4 bit and phased-like 8 bit single stream
    d = values((128, 1), False, 3.)
    pl = gsb.GSBPayload.fromdata(d, bps=4)
    pl.words = pl.words.copy()
    run('gsb_r4', pl, lambda o: [o.words], [[3], [S(5, 17)], [S(2, 40, 3)], [S(100, None)]], False, 3.,
        dict(kind='gsb_payload', bps=4, complex_data=False, sample_shape=[1]))
    # VDIF frame set: 4 threads, 2 channels, 2 bit real
    d = values((32, 4, 2), False, 2.)
    h0 = vdif.VDIFHeader.fromvalues(edv=0, time=Time('2015-06-07T08:09:10'), nchan=2, bps=2,
                                    complex_data=False, thread_id=0, samples_per_frame=32, station='AA')
    fs = vdif.VDIFFrameSet.fromdata(d, h0)
    for f in fs.frames:
        f.payload.words = f.payload.words.copy()
    run('vdif_fs', fs, lambda o: [f.payload.words for f in o.frames],
        [[S(3, 9), 1], [S(None, None), 2, 1], [5], [S(0, 32, 5), S(1, 3), 0], [7, 3, 1], [S(None, None)]],
        False, 2., dict(kind='vdif_frameset', nthread=4, nchan=2, bps=2, samples_per_frame=32,
                        words=[int(w) for w in h0.words]))
    np.savez_compressed(os.path.join(GOLD, 'setitem_cases.npz'), **arrays)
    with open(os.path.join(GOLD, 'setitem_cases.json'), 'w') as f:
        json.dump(meta, f, indent=0)
    print('setitem:', [(m['name'], sum('error' in o for o in m['ops'])) for m in meta])


def gen_utils():
    """Known answers of the reference's helpers (base/utils.py:13-250): BCD,
    CRC per value (Mark 5B polynomial and a short one), CRC over stacked bit
    streams (Mark 4 polynomial), byte patterns."""
    from baseband.base.utils import bcd_decode, bcd_encode, CRC, CRCStack, byte_array, lcm
    rng = np.random.default_rng(99)
    out = {}
    vals = [int(v) for v in rng.integers(0, 10**8, 20)]
    out['bcd'] = [[v, int(bcd_encode(v))] for v in vals]
    arr = rng.integers(0, 10**8, 16).astype(np.uint32)
    out['bcd_array'] = [arr.tolist(), np.asarray(bcd_encode(arr)).tolist()]
    out['crc'] = []
    for pol in (0x18005, 0x180f, 0x13):
        c = CRC(pol)
        ints = [int(v) for v in rng.integers(0, 2**48, 10)] + [0, 1, 2**80 + 12345]
        out['crc'].append(dict(polynomial=pol, length=len(c), values=[[str(v), int(c(v))] for v in ints],
                               array=[[int(v) for v in ints[:10]],
                                      np.asarray(c(np.array(ints[:10], dtype='u8'))).tolist()]))
    cs = CRCStack(0x180f)
    stream = rng.integers(0, 2**32, 148).astype(np.uint32)
    bits = rng.integers(0, 2, 60).astype(bool)
    out['crcstack'] = dict(stream=stream.tolist(), crc=np.asarray(cs(stream)).tolist(),
                           bits=bits.astype(int).tolist(), bits_crc=np.asarray(cs(bits)).astype(int).tolist())
    out['byte_array'] = [[[0xabaddeed], byte_array(0xabaddeed).tolist()],
                         [[1, 2**32 - 1], byte_array([1, 2**32 - 1]).tolist()]]
    out['lcm'] = [[4, 6, int(lcm(4, 6))], [21, 6, int(lcm(21, 6))]]
    with open(os.path.join(GOLD, 'utils_cases.json'), 'w') as f:
        json.dump(out, f)
    print('utils:', {k: len(v) for k, v in out.items()})


def gen_block_writers():
    """DADA / GUPPI stream writers of the reference (dada/base.py:333-362,
    guppi/base.py:281-310) on seeded non-integer data (exercises the round +
    clip of the int8 encoders), including a padded partial last frame.  Stored:
    the input samples and the bytes of the file the reference wrote."""
    import warnings
    out = {}
    tmp = tempfile.mkdtemp()
    t0 = Time('2019-03-01T12:00:00', precision=9)
    rng = np.random.default_rng(2024)

    def rnd(shape, cplx):
        x = rng.standard_normal(shape) * 60.
        if cplx:
            x = x + 1j * rng.standard_normal(shape) * 60.
        return x.astype(np.complex64 if cplx else np.float32)

    # DADA: 2 pol, 1 channel, complex; 3.4 frames of 500 samples
    h = dada.DADAHeader.fromvalues(time=t0, sample_rate=16 * u.MHz, samples_per_frame=500,
                                   bps=8, complex_data=True, npol=2, nchan=1,
                                   telescope='TEST', instrument='bbamd')
    data = rnd((1700, 2), True)
    f = os.path.join(tmp, 'w.dada')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with dada.open(f, 'ws', header0=h) as fw:
            fw.write(data[:123])
            fw.write(data[123:])
    out['dada_in'] = data
    out['dada_file'] = np.fromfile(f, np.uint8)
    with dada.open(f, 'rs') as fr:
        out['dada_back'] = fr.read()
    # DADA real, 4 channels
    h = dada.DADAHeader.fromvalues(time=t0, sample_rate=1 * u.MHz, samples_per_frame=256,
                                   bps=8, complex_data=False, npol=1, nchan=4)
    data = rnd((512, 4), False)
    f = os.path.join(tmp, 'r.dada')
    with dada.open(f, 'ws', header0=h) as fw:
        fw.write(data)
    out['dada_real_in'] = data
    out['dada_real_file'] = np.fromfile(f, np.uint8)
    # GUPPI channels-first and time-first, overlap 0
    for key, cf, nchan, npol, spf, n in (('guppi_cf', True, 8, 2, 128, 3 * 128),
                                         ('guppi_tf', False, 4, 2, 64, 2 * 64 + 20)):
        bpcs = nchan * npol * 2
        h = guppi.GUPPIHeader.fromvalues(time=t0, sample_rate=1 * u.MHz, samples_per_frame=spf,
                                         overlap=0, npol=npol, nchan=nchan, pktsize=spf * bpcs // 4,
                                         bps=8, pktfmt=('1SFA' if cf else 'SIMPLE'))
        data = rnd((n, npol, nchan), True)
        f = os.path.join(tmp, key + '.raw')
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            with guppi.open(f, 'ws', header0=h) as fw:
                fw.write(data[:50])
                fw.write(data[50:])
        out[key + '_in'] = data
        out[key + '_file'] = np.fromfile(f, np.uint8)
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(GOLD, 'block_writer_cases.npz'), **out)
    print('block writers:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    which = sys.argv[1:] or ['all']
    steps = [('levels', gen_levels), ('vdif_samples', gen_vdif_samples),
             ('mark5b_sample', gen_mark5b_sample), ('vdif_synth', gen_vdif_synth),
             ('vdif_invalid', gen_vdif_invalid), ('mark5b_synth', gen_mark5b_synth),
             ('mark4_bitmaps', gen_mark4_bitmaps), ('mark4_samples', gen_mark4_samples),
             ('mark4_synth', gen_mark4_synth), ('guppi', gen_guppi), ('dada', gen_dada),
             ('gsb', gen_gsb), ('vdif_corrupt', gen_vdif_corrupt),
             ('vdif_edv_ab', gen_vdif_edv_ab), ('encode', gen_encode),
             ('sequence', gen_sequence), ('block_writers', gen_block_writers),
             ('fixed_corrupt', gen_fixed_corrupt), ('info', gen_info), ('gsb_writer', gen_gsb_writer),
             ('stream_fuzz', gen_stream_fuzz), ('item_fuzz', gen_item_fuzz), ('locate', gen_locate), ('header_fuzz', gen_header_fuzz),
             ('mark4_header', gen_mark4_header), ('setitem', gen_setitem), ('utils', gen_utils)]
    mpath = os.path.join(GOLD, 'manifest.json')
    if os.path.exists(mpath) and which != ['all']:
        with open(mpath) as f:
            manifest = json.load(f)
    for nm, fn in steps:
        if 'all' in which or nm in which:
            fn()
    with open(mpath, 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print('wrote', mpath, len(manifest['cases']), 'cases')
