"""CPU oracle (NumPy) for the baseband decode hot path -- TEST INFRASTRUCTURE.

This module is a from-scratch NumPy restatement of the algorithms of the
reference package mhvk/baseband (pure Python/NumPy, GPLv3) for the path
``open().read() -> Payload.fromfile -> Payload.data``.  It is the checker the
HIP kernels are compared against; it is never imported by the product package
``baseband_amd`` (only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it).

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference in
the development container and (a) checks every function below against it on
seeded inputs and on the reference's own sample files, (b) writes the golden
vectors under ``tests/golden/`` that ``tests/test_oracle_*.py`` re-check
everywhere.  Rows without a reference counterpart (Mark 4 "longitudinal
parity", DADA NBIT=32) are not implemented here.

Every function cites the reference lines it follows (paths relative to the
reference tree).
"""
import numpy as np

# --------------------------------------------------------------------------
# Levels (base/encoding.py:14,46-56)
# --------------------------------------------------------------------------
OPTIMAL_2BIT_HIGH = 3.316505
TWO_BIT_1_SIGMA = 2.174564
FOUR_BIT_1_SIGMA = 2.95
EIGHT_BIT_1_SIGMA = 71.0 / 2.

LEVELS_1 = np.array([-1.0, 1.0], dtype=np.float32)
LEVELS_2 = np.array([-OPTIMAL_2BIT_HIGH, -1.0, 1.0, OPTIMAL_2BIT_HIGH],
                    dtype=np.float32)
# float32 division, not a reciprocal multiply (base/encoding.py:56)
LEVELS_4 = ((np.arange(16, dtype=np.float32) - np.float32(8.))
            / np.float32(FOUR_BIT_1_SIGMA)).astype(np.float32)


def levels_8bit_vdif():
    """256-entry table of base/encoding.py:131-144 (subtract, then divide)."""
    b = np.arange(256, dtype=np.uint8).astype(np.float32)
    b -= np.float32(127.5)
    b /= np.float32(EIGHT_BIT_1_SIGMA)
    return b


def code_levels(coder, bps):
    """Code -> float32 level table, 2**bps entries.

    coder 'vdif'   : offset binary (vdif/payload.py:53-63)
    coder 'mark5b' : sign/magnitude, field f = s + 2 m, level index 2 s + m
                     (mark5b/payload.py:60-66); 1 bit: set -> -1
    coder 'int'    : two's complement (gsb/payload.py:24-42, dada/payload.py:13-14)
    """
    if coder == 'vdif':
        return {1: LEVELS_1, 2: LEVELS_2, 4: LEVELS_4,
                8: levels_8bit_vdif()}[bps]
    if coder == 'mark5b':
        if bps == 1:
            return LEVELS_1[[1, 0]]
        if bps == 2:
            f = np.arange(4)
            return LEVELS_2[2 * (f & 1) + (f >> 1)]
        raise KeyError(bps)
    if coder == 'int':
        if bps == 4:
            n = np.arange(16)
            return np.where(n < 8, n, n - 16).astype(np.float32)
        if bps == 8:
            return np.arange(256, dtype=np.uint8).view(np.int8).astype(np.float32)
        raise KeyError(bps)
    raise KeyError(coder)


def byte_lut(coder, bps):
    """256 x (8/bps) table: byte value -> samples in time order (LSB first).

    Same construction as vdif/payload.py:53-63 / mark5b/payload.py:60-72.
    """
    lev = code_levels(coder, bps)
    b = np.arange(256)[:, np.newaxis]
    i = np.arange(0, 8, bps)
    return lev[(b >> i) & ((1 << bps) - 1)]


_LUT_CACHE = {}


def decode_flat(raw, coder, bps):
    """Flat decode of payload bytes: lut.take(words.view(u1), axis=0).

    vdif/payload.py:69-103, mark5b/payload.py:78-94, base/encoding.py:131-144,
    gsb/payload.py:24-42, dada/payload.py:13-14.  Returns 1-D float32.
    """
    b = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) \
        else raw.view(np.uint8).ravel()
    if bps == 8 and coder == 'int':
        return b.view(np.int8).astype(np.float32)
    key = (coder, bps)
    if key not in _LUT_CACHE:
        _LUT_CACHE[key] = byte_lut(coder, bps)
    return _LUT_CACHE[key].take(b, axis=0).ravel()


def payload_data(raw, coder, bps, sample_shape, complex_data):
    """PayloadBase._decode + __getitem__(()) (base/payload.py:314-330)."""
    flat = decode_flat(raw, coder, bps)
    if complex_data:
        flat = flat.view(np.complex64)
    return flat.reshape((-1,) + tuple(sample_shape))


# --------------------------------------------------------------------------
# VDIF headers (vdif/header.py:529-542,557-559,595-598; base/header.py:35-87)
# --------------------------------------------------------------------------
def vdif_header_fields(w):
    """Extract the fields of an 8-word (or 4-word legacy) VDIF header."""
    w = [int(x) for x in w]
    h = dict(
        invalid_data=bool(w[0] >> 31 & 1),
        legacy_mode=bool(w[0] >> 30 & 1),
        seconds=w[0] & 0x3fffffff,
        ref_epoch=w[1] >> 24 & 0x3f,
        frame_nr=w[1] & 0xffffff,
        vdif_version=w[2] >> 29 & 0x7,
        lg2_nchan=w[2] >> 24 & 0x1f,
        frame_length=w[2] & 0xffffff,
        complex_data=bool(w[3] >> 31 & 1),
        bits_per_sample=w[3] >> 26 & 0x1f,
        thread_id=w[3] >> 16 & 0x3ff,
        station_id=w[3] & 0xffff)
    h['edv'] = False if h['legacy_mode'] else (w[4] >> 24 & 0xff)
    h['header_nbytes'] = 16 if h['legacy_mode'] else 32
    h['frame_nbytes'] = h['frame_length'] * 8          # vdif/header.py:293-296
    h['payload_nbytes'] = h['frame_nbytes'] - h['header_nbytes']
    h['bps'] = h['bits_per_sample'] + 1                # :313-315
    h['nchan'] = 2 ** h['lg2_nchan']                   # :336-338
    # vdif/header.py:359-364
    vpw = 32 // h['bps'] // (2 if h['complex_data'] else 1)
    h['samples_per_frame'] = h['payload_nbytes'] // 4 * vpw // h['nchan']
    if h['edv'] in (1, 3):                             # :595-598, :610-619
        h['sampling_unit'] = bool(w[4] >> 23 & 1)
        h['sampling_rate'] = w[4] & 0x7fffff
        h['sync_pattern'] = w[5]
    return h


def vdif_verify(h, w):
    """VDIFBaseHeader.verify and subclasses (vdif/header.py:550-589,735-737)."""
    if h['legacy_mode']:
        assert h['frame_length'] >= 2
        return
    assert h['frame_length'] >= 4
    if h['edv'] == 0:
        assert all(int(x) == 0 for x in w[4:8])
    if h['edv'] in (1, 3):
        assert h['sync_pattern'] == 0xACABFEED
    if h['edv'] == 3:
        assert h['frame_length'] in (129, 629)


def vdif_stream_mask(edv):
    """Stream-invariant mask words (base/header.py:588-638 with the keys of
    vdif/header.py:109-111,560-566,603-605,727-732)."""
    m = [0] * 8
    m[0] |= 1 << 30                     # legacy_mode
    m[2] |= 0x7 << 29                   # vdif_version
    m[2] |= 0x1f << 24                  # lg2_nchan
    m[2] |= 0xffffff                    # frame_length
    m[3] |= 1 << 31                     # complex_data
    m[3] |= 0x1f << 26                  # bits_per_sample
    m[3] |= 0xffff                      # station_id
    if edv is False:
        return m[:4]
    m[4] |= 0xff << 24                  # edv
    if edv in (1, 3):
        m[4] |= 1 << 23                 # sampling_unit
        m[4] |= 0x7fffff                # sampling_rate
        m[5] |= 0xffffffff              # sync_pattern
    if edv == 3:
        m[7] |= 0xf << 12               # major_rev
        m[7] |= 0xf << 8                # minor_rev
        m[7] |= 0xff                    # personality
    if edv == 2:
        m[4] |= 0xfffff << 4            # sync_pattern (4, 4, 20)
    if edv == 0xab:
        m[2] |= 0xffffff                # frame_length
        m[4] |= 0xffffffff              # Mark 5B sync word lives in word 4
    return m


def vdif_frame_rate(headers):
    """Frames per second from the frame numbers seen in one second
    (base/base.py:371-406): max frame_nr + 1 where the count wraps."""
    h0 = headers[0]
    mx = 0
    for h in headers:
        if h['seconds'] != h0['seconds'] and h['frame_nr'] == 0 and mx > 0:
            break
        mx = max(mx, h['frame_nr'])
    return mx + 1


def vdif_sample_rate_from_header(h):
    """EDV 1/3 sample rate in Hz (vdif/header.py:610-619); None otherwise."""
    if h['edv'] in (1, 3) and h.get('sampling_rate', 0):
        return (h['sampling_rate'] * (1 if h['complex_data'] else 2)
                * (1000000 if h['sampling_unit'] else 1000))
    return None


def vdif_read(raw, frame_rate=None, fill_value=0., thread_ids=None,
              coder=None):
    """Reference-as-written VDIF stream decode of a clean file image.

    Follows VDIFStreamReader.__init__ / get_thread_ids (vdif/base.py:172-215,
    441-457), StreamReaderBase.read (base/base.py:919-969), the frameset
    gather (vdif/frame.py:176-243,402-434), the validity fill
    (base/frame.py:191-199) and the LUT decode, one Python iteration per
    frameset -- this loop IS the reference algorithm and is what the CPU
    baseline times.

    Returns (out, info) with out of shape (nsample, nthread, nchan).
    """
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    w0 = buf[:32].view('<u4')
    h0 = vdif_header_fields(w0)
    vdif_verify(h0, w0)
    fn, hn, pn = h0['frame_nbytes'], h0['header_nbytes'], h0['payload_nbytes']
    nframes_file = len(buf) // fn
    # thread census over leading framesets (vdif/base.py:172-215)
    hdr_words = np.lib.stride_tricks.as_strided(
        buf[:nframes_file * fn].view('<u4'), shape=(nframes_file, hn // 4),
        strides=(fn, 4))
    all_threads = (hdr_words[:, 3] >> 16) & 0x3ff
    frame_nrs = hdr_words[:, 1] & 0xffffff
    seconds = hdr_words[:, 0] & 0x3fffffff
    seen, n_check, k = set(), 1, 0
    while n_check > 0 and k < nframes_file:
        fnr, n0 = frame_nrs[k], len(seen)
        while k < nframes_file and frame_nrs[k] == fnr:
            seen.add(int(all_threads[k]))
            k += 1
        n_check = 2 if len(seen) > n0 else n_check - 1
    file_threads = sorted(seen)
    nthread_file = len(file_threads)
    if thread_ids is None:
        thread_ids = file_threads
    slot = {t: i for i, t in enumerate(thread_ids)}
    if frame_rate is None:
        sr = vdif_sample_rate_from_header(h0)
        if sr is not None:
            frame_rate = int(round(sr / h0['samples_per_frame']))
        else:
            same = seconds == seconds[0]
            frame_rate = int(frame_nrs[same].max()) + 1
    if coder is None:
        coder = 'mark5b' if h0['edv'] == 0xab else 'vdif'
    bps, nchan, cplx = h0['bps'], h0['nchan'], h0['complex_data']
    spf = h0['samples_per_frame']
    # last header of header0's thread (vdif/base.py:492-517)
    same_thread = np.nonzero(all_threads == h0['thread_id'])[0]
    last = same_thread[-1]
    idx_last = int((int(seconds[last]) - h0['seconds']) * frame_rate
                   + int(frame_nrs[last]) - h0['frame_nr'])
    nset = idx_last + 1
    dtype = np.complex64 if cplx else np.float32
    out = np.empty((nset * spf, len(thread_ids), nchan), dtype=dtype)
    set_nbytes = fn * nthread_file
    lut = byte_lut(coder, bps) if not (coder == 'int' and bps == 8) else None
    for i in range(nset):                      # base/base.py:957-967
        base = i * set_nbytes
        for k in range(nthread_file):          # vdif/frame.py:190-226
            o = base + k * fn
            w = buf[o:o + hn].view('<u4')
            tid = int(w[3] >> 16) & 0x3ff
            if tid not in slot:
                continue
            # A set is the run of frames that share the frame_nr of its first header
            # (VDIFFrameSet.fromfile, vdif/frame.py:201-216: "we cannot always rely on
            # header['seconds']"), and it is that first header whose index is checked
            # (frameset['seconds'] is header0's, vdif/frame.py:243; base/base.py:1108-1110).
            wl = buf[base:base + hn].view('<u4')
            assert int(w[1] & 0xffffff) == int(wl[1] & 0xffffff), "could not find all requested frames."
            idx = int((int(wl[0] & 0x3fffffff) - h0['seconds']) * frame_rate
                      + int(wl[1] & 0xffffff) - h0['frame_nr'])
            assert idx == i, "wrong frame number"
            if w[0] >> 31:                     # base/frame.py:191-199
                out[i * spf:(i + 1) * spf, slot[tid]] = fill_value
                continue
            words = buf[o + hn:o + fn]
            d = lut.take(words, axis=0).ravel()          # vdif/payload.py:83-86
            if cplx:
                d = d.view(np.complex64)
            out[i * spf:(i + 1) * spf, slot[tid]] = d.reshape(-1, nchan)[:spf]
    info = dict(header0=h0, thread_ids=list(thread_ids), frame_rate=frame_rate,
                samples_per_frame=spf, nframesets=nset)
    return out, info


# --------------------------------------------------------------------------
# Mark 5B (mark5b/header.py:60-68,177-185; frame.py:62-70; base.py:206-213)
# --------------------------------------------------------------------------
MARK5B_SYNC = 0xABADDEED
MARK5B_FRAME_NBYTES = 10016
MARK5B_PAYLOAD_NBYTES = 10000
MARK5B_FILL = 0x11223344


def bcd_decode(v, ndigit):
    """base/utils.py:18-40 restricted to ndigit digits."""
    r, m = 0, 1
    for i in range(ndigit):
        d = (v >> (4 * i)) & 0xf
        if d > 9:
            raise ValueError("invalid BCD encoded value")
        r += d * m
        m *= 10
    return r


def mark5b_header_fields(w):
    w = [int(x) for x in w]
    return dict(sync_pattern=w[0], user=w[1] >> 16 & 0xffff,
                internal_tvg=bool(w[1] >> 15 & 1), frame_nr=w[1] & 0x7fff,
                bcd_jday=w[2] >> 20 & 0xfff, bcd_seconds=w[2] & 0xfffff,
                bcd_fraction=w[3] >> 16 & 0xffff, crc=w[3] & 0xffff,
                jday=bcd_decode(w[2] >> 20 & 0xfff, 3),
                seconds=bcd_decode(w[2] & 0xfffff, 5))


def mark5b_read(raw, nchan, bps=2, frame_rate=None, fill_value=0.):
    """Reference-as-written Mark 5B decode of a clean file image
    (mark5b/base.py:228-301; mark5b/frame.py:62-70; payload.py:78-94)."""
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    fn = MARK5B_FRAME_NBYTES
    nframes = len(buf) // fn
    h0 = mark5b_header_fields(buf[:16].view('<u4'))
    assert h0['sync_pattern'] == MARK5B_SYNC
    hdrs = np.lib.stride_tricks.as_strided(
        buf[:nframes * fn].view('<u4'), shape=(nframes, 4), strides=(fn, 4))
    if frame_rate is None:
        same = hdrs[:, 2] == hdrs[0, 2]
        frame_rate = int((hdrs[same, 1] & 0x7fff).max()) + 1
    spf = MARK5B_PAYLOAD_NBYTES * 8 // bps // nchan
    hl = mark5b_header_fields(hdrs[nframes - 1])
    idx_last = ((hl['jday'] - h0['jday']) * 86400 + hl['seconds'] - h0['seconds']) \
        * frame_rate + hl['frame_nr'] - h0['frame_nr']
    nfr = idx_last + 1
    out = np.empty((nfr * spf, nchan), dtype=np.float32)
    lut = byte_lut('mark5b', bps)
    for i in range(nfr):
        o = i * fn
        w = buf[o:o + 16].view('<u4')
        assert w[0] == MARK5B_SYNC
        words = buf[o + 16:o + fn]
        if np.all(words.view('<u4') == MARK5B_FILL):     # mark5b/frame.py:62-70
            out[i * spf:(i + 1) * spf] = fill_value
            continue
        out[i * spf:(i + 1) * spf] = lut.take(words, axis=0).reshape(-1, nchan)
    return out, dict(header0=h0, frame_rate=frame_rate, samples_per_frame=spf,
                     nframes=nfr)


# --------------------------------------------------------------------------
# Mark 4 (mark4/header.py:26-64,107-143,306-328,540-650; payload.py:48-342;
#         frame.py:78-97,148-263)
# --------------------------------------------------------------------------
MARK4_DTYPES = {8: '<u1', 16: '<u2', 32: '<u4', 64: '<u8'}
MARK4_FT_SIGNATURE = 0xf0faf050f0faf05


def mark4_stream2words(stream, track=None):
    """Track transpose: bit `track` of stream word 32 j + i becomes bit
    31 - i of header word j (mark4/header.py:47-64)."""
    stream = np.asarray(stream)
    if track is None:
        track = np.arange(stream.dtype.itemsize * 8, dtype=stream.dtype)
    sel = ((stream.reshape(-1, 32, 1) >> track) & 1).astype(np.uint32)
    sel <<= np.arange(31, -1, -1, dtype=np.uint32).reshape(-1, 1)
    return np.bitwise_or.reduce(sel, axis=1)


def mark4_header_fields(words):
    """Per-track fields of a (5, ntrack) header word array
    (mark4/header.py:107-143)."""
    w = np.asarray(words, dtype=np.uint32)
    f = dict(
        fan_out=(w[1] >> 22) & 0x3, magnitude_bit=((w[1] >> 21) & 1).astype(bool),
        lsb_output=((w[1] >> 20) & 1).astype(bool), converter_id=(w[1] >> 16) & 0xf,
        time_sync_error=((w[1] >> 15) & 1).astype(bool),
        internal_clock_error=((w[1] >> 14) & 1).astype(bool),
        processor_time_out_error=((w[1] >> 13) & 1).astype(bool),
        communication_error=((w[1] >> 12) & 1).astype(bool),
        system_id=w[1] & 0xff, sync_pattern=w[2],
        bcd_unit_year=(w[3] >> 28) & 0xf, bcd_day=(w[3] >> 16) & 0xfff,
        bcd_hour=(w[3] >> 8) & 0xff, bcd_minute=w[3] & 0xff,
        bcd_second=(w[4] >> 24) & 0xff, bcd_fraction=(w[4] >> 12) & 0xfff,
        crc=w[4] & 0xfff)
    ntrack = w.shape[1]
    f['ntrack'] = ntrack
    f['fanout'] = int(f['fan_out'].max()) + 1                  # :565-571
    f['bps'] = 2 if f['magnitude_bit'].any() else 1             # :611-617
    f['nchan'] = ntrack // (f['fanout'] * f['bps'])             # :636-641
    f['frame_nbytes'] = ntrack * 2500                           # :550-553
    f['header_nbytes'] = ntrack * 20                            # :545-548
    f['samples_per_frame'] = 20000 * f['fanout']                # :584-592
    f['valid'] = not np.any(f['time_sync_error'] | f['internal_clock_error']
                            | f['processor_time_out_error']
                            | f['communication_error'])         # frame.py:78-87
    return f


def mark4_time_quarter_ms(f, track=0):
    """Header time of one track in units of 0.25 ms since the start of the
    (unit) year: day, hour, minute, second BCD + millisecond BCD whose last
    digit d encodes d*1.25 ms (mark4/header.py:198-214,223-241)."""
    day = bcd_decode(int(f['bcd_day'][track]), 3)
    hour = bcd_decode(int(f['bcd_hour'][track]), 2)
    minute = bcd_decode(int(f['bcd_minute'][track]), 2)
    sec = bcd_decode(int(f['bcd_second'][track]), 2)
    ms = bcd_decode(int(f['bcd_fraction'][track]), 3)
    q = 4 * ms + ms % 5
    return (((day * 24 + hour) * 60 + minute) * 60 + sec) * 4000 + q


def _m4_reorder32(x):
    """mark4/payload.py:48-52."""
    x = x.astype(np.uint32)
    return ((x & np.uint32(0xAA55AA55)) | ((x & np.uint32(0x55005500)) >> np.uint32(7))
            | ((x & np.uint32(0x00AA00AA)) << np.uint32(7)))


def _m4_reorder64(x):
    """mark4/payload.py:56-60."""
    x = x.astype(np.uint64)
    return ((x & np.uint64(0xAA55AA55AA55AA55))
            | ((x & np.uint64(0x5500550055005500)) >> np.uint64(7))
            | ((x & np.uint64(0x00AA00AA00AA00AA)) << np.uint64(7)))


def _m4_reorder64_ft(x):
    """mark4/payload.py:62-69."""
    x = x.astype(np.uint64)
    return ((x & np.uint64(0xFFFFFAAFFFFFFAAF))
            | ((x & np.uint64(0x0000050000000500)) >> np.uint64(4))
            | ((x & np.uint64(0x0000005000000050)) << np.uint64(4)))


def _m4_luts():
    """mark4/payload.py:88-115: byte -> 4 samples for the three sign/mag
    placements."""
    b = np.arange(256)[:, np.newaxis]
    i = np.arange(4)
    out = []
    for s, m in ((i * 2, i * 2 + 1), (i + (i // 2) * 2, i + (i // 2) * 2 + 2),
                 (i, i + 4)):
        out.append(LEVELS_2[2 * (b >> s & 1) + (b >> m & 1)])
    return out


_M4_LUT1, _M4_LUT2, _M4_LUT3 = _m4_luts()


def mark4_decode(words, nchan, fanout, magnitude_signature=None):
    """The five registered Mark 4 decoders (mark4/payload.py:122-288,333-342).
    `words` has the stream dtype of its track count."""
    words = np.ascontiguousarray(words)
    key = (nchan, magnitude_signature if magnitude_signature is not None else 2, fanout)
    if key == (2, 2, 4):                       # 16 tracks
        fr = words.view(np.uint8).reshape(-1, 2)
        return _M4_LUT3.take(fr, axis=0).transpose(1, 0, 2).reshape(2, -1).T
    if key == (4, 2, 4):                       # 32 tracks
        fr = _m4_reorder32(words.view('<u4')).view(np.uint8).reshape(-1, 4)
        fr = fr.take(np.array([0, 2, 1, 3]), axis=1)
        return _M4_LUT1.take(fr.T, axis=0).reshape(4, -1).T
    if key == (8, 2, 2):                       # 32 tracks
        fr = words.view(np.uint8).reshape(-1, 4)
        return (_M4_LUT3.take(fr, axis=0).reshape(-1, 4, 2, 2)
                .transpose(3, 1, 0, 2).reshape(8, -1).T)
    if key == (8, 2, 4):                       # 64 tracks
        fr = _m4_reorder64(words.view('<u8')).view(np.uint8).reshape(-1, 8)
        fr = fr.take(np.array([0, 2, 1, 3, 4, 6, 5, 7]), axis=1)
        return _M4_LUT1.take(fr.T, axis=0).reshape(8, -1).T
    if key == (16, MARK4_FT_SIGNATURE, 2):     # 64 tracks, Fortaleza
        fr = _m4_reorder64_ft(words.view('<u8')).view(np.uint8).reshape(-1, 8)
        return (_M4_LUT3.take(fr, axis=0).reshape(-1, 2, 4, 2, 2)
                .transpose(1, 4, 2, 0, 3).reshape(16, -1).T)
    raise KeyError(key)


# Track assignments, tables 10-14 of the Mark 4 memo 230.3, as tabulated in
# mark4/header.py:306-328; shape (fanout, nchan, bps) for 32 tracks.
_M4_TA = {
    (2, 4): np.array([[2, 10, 3, 11, 18, 26, 19, 27], [4, 12, 5, 13, 20, 28, 21, 29],
                      [6, 14, 7, 15, 22, 30, 23, 31], [8, 16, 9, 17, 24, 32, 25, 33]]
                     ).reshape(4, 4, 2) - 2,
    (2, 2): np.array([[2, 6, 3, 7, 10, 14, 11, 15, 18, 22, 19, 23, 26, 30, 27, 31],
                      [4, 8, 5, 9, 12, 16, 13, 17, 20, 24, 21, 25, 28, 32, 29, 33]]
                     ).reshape(2, 8, 2) - 2,
    (2, 1): np.array([[2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32,
                       3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 33]]
                     ).reshape(1, 16, 2) - 2,
}


def mark4_track_assignment(ntrack, bps, fanout):
    """mark4/header.py:381-399."""
    ta = _M4_TA[(bps, fanout)]
    if ntrack == 64:
        return np.concatenate((ta, ta + 32), axis=1)
    if ntrack == 32:
        return ta
    if ntrack == 16:
        return ta[:, ::2, :] // 2
    raise ValueError(ntrack)


def mark4_magnitude_signature(f):
    """None for the standard sign/magnitude placement, else the packed
    magnitude bits (mark4/payload.py:346-357)."""
    if f['bps'] == 1:
        return None
    ta = mark4_track_assignment(f['ntrack'], f['bps'], f['fanout'])
    if np.all(f['magnitude_bit'][ta] == [False, True]):
        return None
    return int(np.packbits(f['magnitude_bit']).view(MARK4_DTYPES[f['ntrack']]).item())


def mark4_locate_first_frame(buf, ntrack):
    """First byte offset at which the sync pattern (32 all-ones stream words
    preceded by one all-zero stream word, starting 63 words into the header)
    is found with another one a frame later (mark4/base.py:110-166;
    mark4/header.py:345-373)."""
    dt = np.dtype(MARK4_DTYPES[ntrack])
    isz = dt.itemsize
    fn = ntrack * 2500
    ones = np.iinfo(dt).max
    limit = min(len(buf) - fn - 160 * isz, 2 * fn)

    def sync_at(o):
        w = np.frombuffer(buf[o + 63 * isz:o + 96 * isz].tobytes(), dtype=dt)
        return w[0] == 0 and np.all(w[1:] == ones)
    for o in range(0, max(limit, 0) + 1):
        if sync_at(o) and (o + fn + 96 * isz > len(buf) or sync_at(o + fn)):
            return o
    raise LookupError("no Mark 4 frame found")


def mark4_read(raw, ntrack, frame_rate=None, fill_value=0.):
    """Reference-as-written Mark 4 decode of a clean file image: first 160
    stream words of every frame are the headers and decode to fill_value
    (mark4/frame.py:185-189,248-258); error flags make the frame invalid."""
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    dt = np.dtype(MARK4_DTYPES[ntrack])
    fn = ntrack * 2500
    off0 = mark4_locate_first_frame(buf, ntrack)
    nframes = (len(buf) - off0) // fn
    out = None
    info = {}
    t0 = None
    for i in range(nframes):
        frame = np.frombuffer(buf[off0 + i * fn:off0 + (i + 1) * fn].tobytes(), dtype=dt)
        f = mark4_header_fields(mark4_stream2words(frame[:160]))
        if i == 0:
            sig = mark4_magnitude_signature(f)
            spf = f['samples_per_frame']
            out = np.empty((nframes * spf, f['nchan']), np.float32)
            info = dict(header0=f, samples_per_frame=spf, offset0=off0,
                        signature=sig, nframes=nframes)
            t0 = mark4_time_quarter_ms(f)
        if frame_rate is not None:
            dq = mark4_time_quarter_ms(f) - t0
            assert dq * frame_rate == i * 4000, "wrong frame number"
        rows = out[i * spf:(i + 1) * spf]
        nfill = 160 * f['fanout']
        rows[:nfill] = fill_value
        if f['valid']:
            rows[nfill:] = mark4_decode(frame[160:], f['nchan'], f['fanout'], sig)
        else:
            rows[nfill:] = fill_value
    return out, info


# --------------------------------------------------------------------------
# GUPPI (guppi/header.py:105-143,216-352; payload.py:13-14,52-110;
#        base.py:196-225,270-278)
# --------------------------------------------------------------------------
def _fits_value(text):
    text = text.split('/')[0].strip() if not text.strip().startswith("'") else text.strip()
    if text.startswith("'"):
        return text[1:text.index("'", 1)].rstrip()
    if text in ('T', 'F'):
        return text == 'T'
    try:
        return int(text)
    except ValueError:
        try:
            return float(text)
        except ValueError:
            return text


def guppi_parse_header(buf, offset=0):
    """80-character cards up to END (guppi/header.py:105-143); returns
    (dict, header_nbytes) with DIRECTIO padding to 512 (guppi/header.py:216-224)."""
    cards, pos = {}, offset
    ncards = 0
    while True:
        line = bytes(buf[pos:pos + 80]).decode('ascii')
        if line == '':
            raise EOFError
        pos += 80
        ncards += 1
        if line[:3] == 'END':
            break
        if line[8] == '=':
            cards[line[:8].strip()] = _fits_value(line[9:])
    nbytes = ncards * 80
    if int(cards.get('DIRECTIO', 0)) and nbytes % 512:
        nbytes += 512 - nbytes % 512
    return cards, nbytes


def guppi_geometry(h):
    nchan = int(h['OBSNCHAN'])
    cplx = nchan != 1                                   # guppi/header.py:254-256
    npol = int(h['NPOL']) // (2 if cplx else 1)         # :259-261
    bps = int(h['NBITS'])
    bpcs = nchan * int(h['NPOL']) * bps                 # :287-289
    spf = int(h['BLOCSIZE']) * 8 // bpcs                # :326-328
    return dict(nchan=nchan, complex_data=cplx, npol=npol, bps=bps,
                samples_per_frame=spf, overlap=int(h.get('OVERLAP', 0)),
                channels_first=h.get('PKTFMT', '1SFA') != 'SIMPLE',
                payload_nbytes=int(h['BLOCSIZE']))


def guppi_payload_data(words, g):
    """Whole-payload decode (guppi/payload.py:90-102): int8 -> float32 ->
    complex, then (chan, time, pol) or (time, chan, pol) -> (time, pol, chan)."""
    d = np.asarray(words).view(np.int8).astype(np.float32)
    if g['complex_data']:
        d = d.view(np.complex64)
    if g['channels_first']:
        return d.reshape(g['nchan'], -1).T.reshape(-1, g['npol'], g['nchan'])
    return d.reshape(-1, g['nchan'], g['npol']).transpose(0, 2, 1)


def guppi_read(raw, offset=0, count=None):
    """Reference-as-written GUPPI stream read of `count` samples from sample
    `offset` (guppi/base.py:203-206,270-278 + the loop of base/base.py:957-967).

    The stream advances by spf - OVERLAP samples per frame, but each loop
    iteration takes everything up to the END of the frame it is in (overlap
    tail included), so inside one read() the overlap samples come from the
    earlier frame and the next frame is entered at its sample OVERLAP."""
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    h0, hn = guppi_parse_header(buf)
    g = guppi_geometry(h0)
    fn = hn + g['payload_nbytes']
    nframes = len(buf) // fn
    spf = g['samples_per_frame']
    keep = spf - g['overlap']
    nsample = nframes * keep + g['overlap']
    if count is None:
        count = nsample - offset
    assert offset + count <= nsample
    normal_end = nsample - g['overlap']
    parts, done = [], 0
    cache = {}
    while done < count:
        o = offset + done
        if normal_end <= o < nsample:                   # guppi/base.py:270-278
            index, so = divmod(normal_end - 1, keep)
            so += 1 + o - normal_end
        else:
            index, so = divmod(o, keep)
        if index not in cache:
            fo = index * fn
            _, hni = guppi_parse_header(buf, fo)
            cache = {index: guppi_payload_data(
                buf[fo + hni:fo + hni + g['payload_nbytes']], g)}
        n = min(count - done, spf - so)
        parts.append(cache[index][so:so + n])
        done += n
    shape = (0, g['npol'], g['nchan'])
    out = (np.ascontiguousarray(np.concatenate(parts)) if parts
           else np.empty(shape, np.complex64 if g['complex_data'] else np.float32))
    return out, dict(header0=h0, header_nbytes=hn, frame_nbytes=fn,
                     nframes=nframes, nsample=nsample, **g)


# --------------------------------------------------------------------------
# DADA (dada/header.py:117-136,161-200,289-382; payload.py:13-14,54-89;
#       base.py:251-330)
# --------------------------------------------------------------------------
def dada_parse_header(buf, offset=0):
    """ASCII 'KEY VALUE' lines; default size 4096, HDR_SIZE overrides."""
    hdr_size, pos, h = 4096, offset, {}
    text = bytes(buf[offset:offset + 65536])
    lines = []
    rel = 0
    while rel < hdr_size and text[rel:rel + 1] != b'\x00':
        end = text.index(b'\n', rel) + 1
        line = text[rel:end].decode('ascii')
        rel = end
        if line[0] == '#' and 'end of header' in line:
            break
        if line.startswith('HDR_SIZE'):
            hdr_size = int(line.split()[1])
        lines.append(line)
    for line in lines:
        split = line.strip().split('#')[0].strip().split()
        if len(split) < 2:
            continue
        key, value = split[0], split[1]
        if key in ('FILE_SIZE', 'FILE_NUMBER', 'HDR_SIZE', 'OBS_OFFSET',
                   'OBS_OVERLAP', 'NBIT', 'NDIM', 'NPOL', 'NCHAN',
                   'RESOLUTION', 'DSB'):
            value = int(value)
        elif key in ('FREQ', 'BW', 'TSAMP'):
            value = float(value)
        h[key] = value
    h.setdefault('HDR_SIZE', hdr_size)
    return h


def dada_payload_data(words, h):
    """int8 -> float32 (-> complex); MKBF heaps (heap, pol, chan, 256) ->
    (heap*256, pol, chan) (dada/payload.py:76-79)."""
    npol, nchan, ndim = h['NPOL'], h['NCHAN'], h['NDIM']
    d = np.asarray(words).view(np.int8)
    if h.get('INSTRUMENT') == 'MKBF':
        d = np.moveaxis(d.reshape(-1, npol, nchan, 256, ndim), 3, 1)
    d = np.ascontiguousarray(d).astype(np.float32)
    if ndim == 2:
        d = d.view(np.complex64)
    return d.reshape(-1, npol, nchan)


def dada_read(raw):
    """Frames of HDR_SIZE + FILE_SIZE bytes; a truncated last frame is cut
    to whole payload blocks (dada/base.py:251-306)."""
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    h0 = dada_parse_header(buf)
    hn, pn = h0['HDR_SIZE'], h0['FILE_SIZE']
    fn = hn + pn
    nframes, partial = divmod(len(buf), fn)
    sample_nbytes = h0['NBIT'] * h0['NDIM'] * h0['NPOL'] * h0['NCHAN'] // 8
    block = np.lcm(4, sample_nbytes)
    parts = []
    for i in range(nframes):
        parts.append(dada_payload_data(buf[i * fn + hn:(i + 1) * fn], h0))
    if partial > hn:
        n = (partial - hn) // block * block
        o = nframes * fn + hn
        parts.append(dada_payload_data(buf[o:o + n], h0))
    out = np.ascontiguousarray(np.concatenate(parts))
    return out, dict(header0=h0, nframes=len(parts))


# --------------------------------------------------------------------------
# GSB (gsb/payload.py:24-42,88-131; gsb/base.py:146-201,373-387)
# --------------------------------------------------------------------------
def gsb_read_rawdump(raw, nframes, payload_nbytes, bps=4, nchan=1):
    """Headerless blocks; 4-bit signed nibbles, low nibble first."""
    buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    out = [decode_flat(buf[i * payload_nbytes:(i + 1) * payload_nbytes], 'int', bps)
           .reshape(-1, nchan) for i in range(nframes)]
    return np.concatenate(out)


def gsb_read_phased(files, nframes, payload_nbytes, nchan=512, bps=8,
                    complex_data=True):
    """`files[p][f]`: F consecutive-in-time parts for each polarisation p;
    a frame is (F, n, P, sample_nbytes) words (gsb/payload.py:115-131)."""
    P, F = len(files), len(files[0])
    sample_nbytes = bps * nchan * (2 if complex_data else 1) // 8
    n = payload_nbytes // sample_nbytes
    out = []
    for i in range(nframes):
        words = np.empty((F, n, P, sample_nbytes), np.uint8)
        for p in range(P):
            for f in range(F):
                b = np.asarray(files[p][f])[i * payload_nbytes:(i + 1) * payload_nbytes]
                words[f, :, p, :] = b.reshape(-1, sample_nbytes)
        d = decode_flat(words.ravel(), 'int', bps)
        if complex_data:
            d = d.view(np.complex64)
        out.append(d.reshape(-1, P, nchan))
    return np.concatenate(out)


# --------------------------------------------------------------------------
# Encoders: base/encoding.py:63-158, vdif/payload.py:77-114,
# mark5b/payload.py:78-106, gsb/payload.py:44-52, dada/payload.py:17-18,
# mark4/payload.py:138-300.  float32 in, packed codes out.
# --------------------------------------------------------------------------
def encode_codes(values, coder, bps):
    """Un-packed integer codes for float32 `values` (flat)."""
    v = np.asarray(values, dtype=np.float32)
    if coder in ('vdif', 'mark5b'):
        if bps == 1:
            if coder == 'vdif':                          # encode_1bit_base
                return (v >= np.float32(0.)).astype(np.uint8)
            return np.signbit(v).astype(np.uint8)        # mark5b/payload.py:84-87
        if bps == 2:                                     # encode_2bit_base
            w = np.clip(v, -1.5 * TWO_BIT_1_SIGMA, 1.5 * TWO_BIT_1_SIGMA)
            w = w + np.float32(2 * TWO_BIT_1_SIGMA)
            c = np.floor_divide(w, np.float32(TWO_BIT_1_SIGMA)).astype(np.uint8)
            if coder == 'mark5b':
                c = np.array([0, 2, 1, 3], np.uint8)[c]
            return c
        if coder == 'vdif' and bps == 4:                 # encode_4bit_base
            w = v * np.float32(FOUR_BIT_1_SIGMA)
            w = w + np.float32(8.5)
            return np.clip(w, 0., 15.).astype(np.uint8)
        if coder == 'vdif' and bps == 8:                 # encode_8bit
            return np.clip(np.rint(v * np.float32(EIGHT_BIT_1_SIGMA) + np.float32(127.5)),
                           0, 255).astype(np.uint8)
    if coder == 'int':
        if bps == 8:
            return np.clip(np.rint(v), -128, 127).astype(np.int8).view(np.uint8)
        if bps == 4:
            return (np.clip(np.around(v), -8, 7).astype(np.int8) & 0xf).astype(np.uint8)
    raise KeyError((coder, bps))


def encode_flat(values, coder, bps):
    """Packed bytes, first sample in the least significant bits."""
    c = encode_codes(values, coder, bps).reshape(-1, 8 // bps).astype(np.uint8)
    shifts = (np.arange(8 // bps) * bps).astype(np.uint8)
    return np.bitwise_or.reduce(c << shifts, axis=-1).astype(np.uint8)


def mark4_header_crc_bad(raw, offset0, ntrack, nframes):
    """Per frame the mask of tracks whose 160 header bits do not divide by the
    Mark 4 CRC-12 polynomial 0x180f -- CRCStack._crc / check
    (base/utils.py:200-248, mark4/header.py:34-44) restated: polynomial long
    division carried out on stream words, all tracks at once."""
    raw = np.ascontiguousarray(raw)
    dt = np.dtype(MARK4_DTYPES[ntrack])
    fn = ntrack * 2500
    ones = np.iinfo(dt).max
    pol = np.array([ones if b == '1' else 0 for b in format(0x180f, 'b')], dtype=dt)
    out = []
    for f in range(nframes):
        start = offset0 + f * fn
        stream = np.frombuffer(raw[start:start + 160 * dt.itemsize].tobytes(), dt).copy()
        for i in range(160 - 12):
            stream[i:i + 13] ^= stream[i] & pol
        out.append(int(np.bitwise_or.reduce(stream[-12:])))
    return out
