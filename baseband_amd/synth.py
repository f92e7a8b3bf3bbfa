"""Synthetic file images (host side) for tests, smoke and bench.

The reference's writers cannot travel to the GPU box, so inputs are made
here.  Header words are built with this package's header classes; the byte
layout is checked byte-for-byte against files written by the reference's own
writers (tests/golden/synth/*.bin, see tests/test_host_logic.py).
"""
import numpy as np

from .vdif.header import VDIFHeader, frame_header_words as vdif_frame_headers
from . import synth_codes as enc


def vdif_file_image(payloads, header0, thread_ids=(0,), frame_rate=100,
                    thread_order=None, invalid=()):
    """Assemble a VDIF file image.

    payloads : uint8 array (nsets, nthread, payload_nbytes) in thread_ids order
    invalid  : iterable of (set, thread position) whose invalid_data bit is set
    Returns a 1-D uint8 array.
    """
    payloads = np.asarray(payloads, dtype=np.uint8)
    nsets, nthread, pn = payloads.shape
    assert nthread == len(thread_ids) and pn == header0.payload_nbytes
    order = list(range(nthread)) if thread_order is None else list(thread_order)
    hw = vdif_frame_headers(header0, nsets, thread_ids, frame_rate, order)
    hn = header0.nbytes
    out = np.empty((nsets, nthread, hn + pn), dtype=np.uint8)
    out[:, :, :hn] = hw.view(np.uint8).reshape(nsets, nthread, hn)
    out[:, :, hn:] = payloads[:, order, :]
    for s, p in invalid:
        out[s, order.index(p), 3] |= 0x80
    return out.reshape(-1)


def encode_vdif_stream(data, header0, frame_rate, thread_ids=None,
                       thread_order=None):
    """Encode (nsample, nthread, nchan) data as a complete VDIF file image
    (what the reference's stream writer would produce)."""
    data = np.asarray(data)
    nsample, nthread, nchan = data.shape
    spf = header0.samples_per_frame
    assert nsample % spf == 0
    nsets = nsample // spf
    if thread_ids is None:
        thread_ids = list(range(nthread))
    comp = enc.components(np.ascontiguousarray(
        data.reshape(nsets, spf, nthread, nchan).transpose(0, 2, 1, 3)))
    bps = header0.bps
    if header0.edv == 0xab:
        packed = enc.encode_mark5b(comp, bps)
    else:
        codes = {1: enc.codes_1bit, 2: enc.codes_2bit, 4: enc.codes_4bit,
                 8: enc.codes_8bit}[bps](comp)
        packed = enc.pack_codes(codes, bps)
    payloads = packed.reshape(nsets, nthread, -1)
    return vdif_file_image(payloads, header0, thread_ids, frame_rate,
                           thread_order)


def random_vdif(seed, nsets, *, nthread=1, nchan=1, bps=2, complex_data=False,
                edv=0, payload_nbytes=8000, frame_rate=1000, station='AA',
                time='2020-01-01T00:00:00', thread_order=None, invalid=()):
    """Seeded random VDIF file image (uniform random payload bytes: every
    code equally likely) plus its header0."""
    kw = dict(bps=bps, nchan=nchan, complex_data=complex_data,
              station=station, time=np.datetime64(time), frame_rate=frame_rate)
    if edv == 3:
        kw['frame_length'] = 629
    else:
        kw['payload_nbytes'] = payload_nbytes
    if edv in (1, 3):
        vpw = 32 // bps // (2 if complex_data else 1)
        spf = (kw.get('payload_nbytes', 5000)) // 4 * vpw // nchan
        kw['sample_rate'] = spf * frame_rate
    header0 = VDIFHeader.fromvalues(edv=edv, **kw)
    rng = np.random.default_rng(seed)
    payloads = rng.integers(0, 256, size=(nsets, nthread, header0.payload_nbytes),
                            dtype=np.uint8)
    image = vdif_file_image(payloads, header0, list(range(nthread)), frame_rate,
                            thread_order, invalid)
    return image, header0


# ---------------------------------------------------------------- Mark 4
def mark4_frame_headers(header0, nframes, frame_rate):
    """Stream-word headers (nframes, 160) for consecutive frames starting at
    header0's time; CRC recomputed per frame like the reference's writer."""
    from .mark4.header import words2stream
    out = np.empty((nframes, 160), dtype=header0.stream_dtype)
    t0 = header0.get_time()
    for i in range(nframes):
        h = header0.copy()
        ns = int(round(i * 1e9 / frame_rate))
        h.set_time(t0 + np.timedelta64(ns, 'ns'))
        h.update_crc()
        out[i] = words2stream(h.words)
    return out


def encode_mark4_stream(data, header0, frame_rate):
    """(nsample, nchan) data -> Mark 4 file image: each frame's first
    160*fanout samples are dropped (overwritten by the header), as the
    reference's writer does (mark4/frame.py:139-141)."""
    encode_mark4 = enc.encode_mark4
    data = np.asarray(data)
    spf = header0.samples_per_frame
    nframes = data.shape[0] // spf
    assert nframes * spf == data.shape[0]
    nfill = 160 * header0.fanout
    hdr = mark4_frame_headers(header0, nframes, frame_rate)
    frames = np.empty((nframes, 20000), dtype=header0.stream_dtype)
    frames[:, :160] = hdr
    for i in range(nframes):
        frames[i, 160:] = encode_mark4(data[i * spf + nfill:(i + 1) * spf], header0)
    return frames.reshape(-1).view(np.uint8)


def random_mark4(seed, nframes, *, ntrack=64, fanout=4, frame_rate=400,
                 time='2015-03-02T04:05:06.25', invalid=(), lead_bytes=0):
    """Seeded random Mark 4 file image (uniform random payload bits) plus its
    header0.  `invalid` lists frames that get an error flag on one track;
    `lead_bytes` random bytes (without a sync pattern) precede the first
    frame, as in real recordings."""
    from .mark4.header import Mark4Header
    header0 = Mark4Header.fromvalues(ntrack, time=np.datetime64(time), bps=2,
                                     fanout=fanout)
    rng = np.random.default_rng(seed)
    dt = header0.stream_dtype
    frames = rng.integers(0, 256, size=(nframes, 20000 * dt.itemsize),
                          dtype=np.uint8).view(dt)
    frames = frames.copy()
    frames[:, :160] = mark4_frame_headers(header0, nframes, frame_rate)
    for f in invalid:
        frames[f, 51] |= dt.type(1 << (f % ntrack))     # communication_error bit
    image = frames.reshape(-1).view(np.uint8)
    if lead_bytes:
        lead = rng.integers(0, 255, size=lead_bytes, dtype=np.uint8)  # never 0xff
        image = np.concatenate([lead, image])
    return image, header0
