"""Thin Python wrappers over the C ABI: torch tensors in, torch tensors out.

PyTorch is used only as the device allocator / stream provider; every
computation is a libbbdecode kernel launched on torch's current HIP stream.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


_gpu_checked = False


def require_gpu():
    global _gpu_checked
    if _gpu_checked:
        return
    if not torch.cuda.is_available():
        raise RuntimeError(
            "baseband_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() "
            "is False and there is no CPU decode path in this package.")
    _gpu_checked = True                 # (the check costs microseconds per call)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(t=None):
    """Handle of torch's current HIP stream on the current device.  The raw
    getter avoids ~8 us of Python in ``torch.cuda.current_stream()`` per call
    (three calls per small read).  `t` is a tensor the launch works on: the
    library launches on the CURRENT device (its level tables are per device),
    so data living on another GPU is refused instead of being addressed from
    the wrong device's stream."""
    dev = torch.cuda.current_device()
    if t is not None and t.device.index is not None and t.device.index != dev:
        raise RuntimeError(
            "tensor lives on cuda:{} but the current device is cuda:{}: wrap the call in "
            "`with torch.cuda.device({}):`".format(t.device.index, dev, t.device.index))
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(dev))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_stream_objects = {}        # (device index, raw handle) -> torch.cuda.Stream


def current_stream(device):
    """``torch.cuda.current_stream(device)`` without its ~5 us of Python: the Stream object
    of a raw handle is remembered (two of these per mid-size read: the arena's reuse
    ordering and the scratch sets' events)."""
    if _raw_stream is None:
        return torch.cuda.current_stream(device)
    index = device.index if isinstance(device, torch.device) else int(device)
    if index is None:
        index = torch.cuda.current_device()
    raw = _raw_stream(index)
    s = _stream_objects.get((index, raw))
    if s is None:
        s = _stream_objects[(index, raw)] = torch.cuda.current_stream(index)
        if len(_stream_objects) > 64:       # (streams come and go: do not keep every one ever seen)
            for k in list(_stream_objects)[:32]:
                del _stream_objects[k]
    return s


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def init():
    require_gpu()
    check(lib.bb_init(), 'bb_init')


def to_device_bytes(raw, device=None):
    """Host bytes-like / ndarray -> 1-D uint8 device tensor (simple upload;
    the chunked pinned pipeline lives in `staging`)."""
    require_gpu()
    if isinstance(raw, torch.Tensor):
        t = raw if raw.dtype == torch.uint8 else raw.view(torch.uint8)
        return t.reshape(-1).to(device or 'cuda')
    a = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) \
        else np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    return torch.from_numpy(np.array(a, copy=True)).to(device or 'cuda')


def _empty_output(nelem, device):
    """float32 output tensor of a decode launch: from the placement arena when
    there is one (placement.empty_output), else torch.empty."""
    from .placement import empty_output
    return empty_output(nelem, torch.float32, device)


class _Target:
    """Output tensor for a decode launch.  The kernels store 16 bytes per
    lane, so the library wants `out` 16-byte aligned; a slice of a larger
    result that starts elsewhere (a read that enters a frame at an odd row) is
    decoded into a temporary and copied into place afterwards."""
    __slots__ = ('want', 'use')

    def __init__(self, out, nelem, device):
        self.want = out
        if out is None:
            self.use = _empty_output(nelem, device)
        elif out.data_ptr() % 16 or not out.is_contiguous():
            self.use = _empty_output(out.numel(), device)
        else:
            self.use = out

    def done(self):
        if self.want is None or self.want is self.use:
            return self.use
        self.want.copy_(self.use.view_as(self.want))
        return self.want


def vdif_scan(dbuf, nframes, frame_nbytes, header_nbytes, pattern, mask,
              ref_seconds, ref_frame_nr, frame_rate, first_offset=0, set_nframes=0):
    """-> int32 tensor (nframes, 4): payload offset lo/hi, time_index,
    thread_id | flags << 16 (the 16-byte bb_frame_rec).  `set_nframes`:
    frames per frame set in file order -- sets are then formed as the
    reference's VDIFFrameSet.fromfile forms them (include/bbdecode.h)."""
    p = _lib.VDIFScanParams()
    p.set_nframes = set_nframes
    p.first_offset = first_offset
    p.frame_nbytes = frame_nbytes
    p.header_nbytes = header_nbytes
    for i in range(8):
        p.pattern[i] = int(pattern[i]) if i < len(pattern) else 0
        p.mask[i] = int(mask[i]) if i < len(mask) else 0
    p.ref_seconds = ref_seconds
    p.ref_frame_nr = ref_frame_nr
    p.frame_rate = frame_rate
    recs = torch.empty((nframes, 4), dtype=torch.int32, device=dbuf.device)
    check(lib.bb_vdif_scan(_ptr(dbuf), dbuf.numel(), C.byref(p), _ptr(recs),
                           nframes, _stream(dbuf)), 'bb_vdif_scan')
    return recs


def _vdif_params(frame_nbytes, header_nbytes, pattern, mask, ref_seconds,
                 ref_frame_nr, frame_rate, first_offset=0):
    p = _lib.VDIFScanParams()
    p.first_offset = first_offset
    p.frame_nbytes = frame_nbytes
    p.header_nbytes = header_nbytes
    for i in range(8):
        p.pattern[i] = int(pattern[i]) if i < len(pattern) else 0
        p.mask[i] = int(mask[i]) if i < len(mask) else 0
    p.ref_seconds = ref_seconds
    p.ref_frame_nr = ref_frame_nr
    p.frame_rate = frame_rate
    return p


_NSCRATCH = 4       # sets of window scratch taking turns when the scan runs on a side stream


def _pow2(n):
    return 1 << max(0, int(n) - 1).bit_length()


class _FrameWindow:
    """Scratch handling shared by the window calls: the scan records and the
    index of a window.  With the scan on a SIDE stream (`scan_stream`: the
    verdict of request k + 1 does not queue behind the decode of request k on
    the caller's stream, include/bbdecode.h) there are `_NSCRATCH` = 4 sets
    taking turns: the decode of request k still reads one set while the scans
    of the next requests fill the others -- a caller can queue a burst of three
    more reads before the host is paced by the GPU -- and before a set is filled
    again the side stream waits for the event recorded behind the decode that
    used it last (16 bytes per frame and 8 per index entry each)."""
    __slots__ = ('recs', 'src', 'fill_value', '_sets', '_turn', '_rotating')

    def _scratch(self, nframes, n, dev, scan_stream=None):
        sets = getattr(self, '_sets', None)
        if sets is None:
            sets = self._sets = [[None, None, None] for _ in range(_NSCRATCH)]
            self._turn = 0
        k = 0
        if scan_stream is not None:
            self._turn = (self._turn + 1) % _NSCRATCH
            k = self._turn
        st = sets[k]
        small = (st[0] is None or st[0].shape[0] < nframes or st[0].device != dev
                 or st[1] is None or st[1].numel() < n or st[1].device != dev)
        if small and scan_stream is None:
            st[0] = torch.empty((max(_pow2(nframes), 64), 4), dtype=torch.int32, device=dev)
            st[1] = torch.empty(max(_pow2(n), 64), dtype=torch.int64, device=dev)
            st[2] = None
        elif small:
            # ALL sets at once (with headroom: they are re-made when a request outgrows
            # them): one allocation, and ONE wait of the side stream for the caller's --
            # torch's allocator hands out memory whose last user may still have work
            # queued on the caller's stream, which is safe there and nowhere else
            # (tools/stress_side_scan.py found a scan overwritten by such work)
            cr, cs = max(_pow2(nframes), 64), max(_pow2(n), 64)
            recs = torch.empty((_NSCRATCH * cr, 4), dtype=torch.int32, device=dev)
            src = torch.empty(_NSCRATCH * cs, dtype=torch.int64, device=dev)
            for j in range(_NSCRATCH):
                sets[j][0], sets[j][1], sets[j][2] = recs[j * cr:(j + 1) * cr], src[j * cs:(j + 1) * cs], None
            scan_stream.wait_stream(torch.cuda.current_stream(dev))
        self.recs, self.src = st[0], st[1]
        if scan_stream is not None and not small and st[2] is not None:
            scan_stream.wait_event(st[2])           # the decode that read this set last
        return st

    def _decode_queued(self, st, scan_stream, dev):
        """Behind a window call: remember where the decode that reads set `st`
        stands -- also for a call that did NOT use the side stream, once the
        sets rotate (a small read between large ones uses set 0 on the caller's
        stream; a later side-stream scan must not refill that set under its
        decode: found by tools/stress_side_scan.py)."""
        if scan_stream is not None or getattr(self, '_rotating', False):
            self._rotating = True
            if st[2] is None:
                st[2] = torch.cuda.Event()
            st[2].record(current_stream(dev))


class VDIFWindow(_FrameWindow):
    """One window of a VDIF stream read -- scan, index, verification, decode --
    as ONE library call (bb_vdif_read_window) with argument blocks that are
    built once per reader: a third of the host time of a small read() went
    into marshalling the same constants for four calls."""
    __slots__ = ('scan', 'dec')

    def __init__(self, frame_nbytes, header_nbytes, pattern, mask, ref_seconds, frame_rate,
                 payload_nbytes, coder, bps, chunk, nslot, complex_data, fill_value):
        self.scan = _vdif_params(frame_nbytes, header_nbytes, pattern, mask, ref_seconds, 0, frame_rate)
        p = self.dec = _lib.DecodeParams()
        p.coder, p.bps, p.chunk, p.nslot = coder, bps, chunk, nslot
        p.payload_nbytes = payload_nbytes
        p.complex_data = int(bool(complex_data))
        self.set_fill(fill_value)
        self.recs = self.src = self._sets = None

    def set_fill(self, fill_value):
        fv = complex(fill_value)
        self.dec.fill_re, self.dec.fill_im = fv.real, fv.imag
        self.fill_value = fill_value

    def run(self, dbuf, ref_frame_nr, nframes, thread_slot, nsets, within, out,
            recs_per_index, nstrict, nbad, verified, scan_stream=None):
        """Launch the window on torch's current stream.  `out`: flat float32
        device tensor; `nbad`: int32[1] device counter, or None for no
        verification; `verified`: raw handle of the event to record behind the
        verification launch, or None; `scan_stream`: torch stream for the scan /
        index / verification launches (needs `verified`; `_FrameWindow`)."""
        dev = dbuf.device
        self.scan.ref_frame_nr = ref_frame_nr
        if not verified:
            scan_stream = None
        st = self._scratch(nframes, nsets * self.dec.nslot, dev, scan_stream)
        tgt = _Target(out, out.numel(), dev)
        nsel = within.numel() if within is not None else 0
        check(lib.bb_vdif_read_window(
            _ptr(dbuf), dbuf.numel(), C.byref(self.scan), nframes, _ptr(thread_slot), nsets,
            C.byref(self.dec), _ptr(within), nsel, _ptr(self.recs), _ptr(self.src),
            _ptr(tgt.use), tgt.use.numel(), recs_per_index, nstrict, _ptr(nbad),
            C.c_void_p(verified) if verified else C.c_void_p(0),
            C.c_void_p(scan_stream.cuda_stream) if scan_stream is not None else C.c_void_p(0),
            _stream(dbuf)), 'bb_vdif_read_window')
        self._decode_queued(st, scan_stream, dev)
        tgt.done()


class Mark5BWindow(_FrameWindow):
    """One window of a Mark 5B stream read as one library call
    (bb_mark5b_read_window); see `VDIFWindow`."""
    __slots__ = ('scan', 'dec')

    def __init__(self, ref_seconds, frame_rate, bps, chunk, fill_value):
        p = self.scan = _lib.Mark5BScanParams()
        p.first_offset, p.ref_seconds, p.frame_rate = 0, ref_seconds, frame_rate
        d = self.dec = _lib.DecodeParams()
        d.coder, d.bps, d.chunk, d.nslot, d.payload_nbytes = _lib.CODER_MARK5B, bps, chunk, 1, 10000
        self.recs = self.src = self._sets = None
        self.set_fill(fill_value)

    def set_fill(self, fill_value):
        fv = complex(fill_value)
        self.dec.fill_re, self.dec.fill_im = fv.real, fv.imag
        self.fill_value = fill_value

    def run(self, dbuf, ref_frame_nr, nframes, n, within, out, nstrict, nbad, verified, scan_stream=None):
        self.scan.ref_frame_nr = ref_frame_nr
        if not verified:
            scan_stream = None
        st = self._scratch(nframes, n, dbuf.device, scan_stream)
        tgt = _Target(out, out.numel(), dbuf.device)
        check(lib.bb_mark5b_read_window(
            _ptr(dbuf), dbuf.numel(), C.byref(self.scan), nframes, n, C.byref(self.dec), _ptr(within),
            within.numel() if within is not None else 0, _ptr(self.recs), _ptr(self.src), _ptr(tgt.use),
            tgt.use.numel(), nstrict, _ptr(nbad), C.c_void_p(verified) if verified else C.c_void_p(0),
            C.c_void_p(scan_stream.cuda_stream) if scan_stream is not None else C.c_void_p(0),
            _stream(dbuf)), 'bb_mark5b_read_window')
        self._decode_queued(st, scan_stream, dbuf.device)
        tgt.done()


class Mark4Window(_FrameWindow):
    """One window of a Mark 4 stream read as one library call
    (bb_mark4_read_window); see `VDIFWindow`."""
    __slots__ = ('scan', 'dec', 'nout', 'frame_qms', 'ref_qms')

    def __init__(self, ntrack, ref_year, ref_qms, frame_qms, nwords, sign_bit, mag_bit, select, fill_words,
                 fill_value):
        p = self.scan = _lib.Mark4ScanParams()
        p.first_offset, p.ntrack, p.ref_year, p.frame_qms = 0, ntrack, ref_year, frame_qms
        self.ref_qms, self.frame_qms = ref_qms, frame_qms
        d = self.dec = _lib.Mark4DecodeParams()
        d.ntrack, d.nwords, d.fill_words = ntrack, nwords, fill_words
        for j, (s, m) in enumerate(zip(sign_bit, mag_bit)):
            d.sign_bit[j] = s
            d.mag_bit[j] = m
        self.nout = len(sign_bit) if select else 0
        self.recs = self.src = self._sets = None
        self.set_fill(fill_value)

    def set_fill(self, fill_value):
        self.dec.fill = float(fill_value)
        self.fill_value = fill_value

    def run(self, dbuf, first, nframes, n, out, nstrict, nbad, verified, scan_stream=None):
        self.scan.ref_qms = self.ref_qms + first * self.frame_qms
        if not verified:
            scan_stream = None
        st = self._scratch(nframes, n, dbuf.device, scan_stream)
        tgt = _Target(out, out.numel(), dbuf.device)
        check(lib.bb_mark4_read_window(
            _ptr(dbuf), dbuf.numel(), C.byref(self.scan), nframes, n, C.byref(self.dec), self.nout,
            _ptr(self.recs), _ptr(self.src), _ptr(tgt.use), tgt.use.numel(), nstrict, _ptr(nbad),
            C.c_void_p(verified) if verified else C.c_void_p(0),
            C.c_void_p(scan_stream.cuda_stream) if scan_stream is not None else C.c_void_p(0),
            _stream(dbuf)), 'bb_mark4_read_window')
        self._decode_queued(st, scan_stream, dbuf.device)
        tgt.done()


def vdif_locate(dbuf, nbytes, frame_nbytes, header_nbytes, pattern, mask):
    """Byte-granular header search -> sorted int64 device tensor of frame
    offsets (corruption-tolerant discovery)."""
    p = _vdif_params(frame_nbytes, header_nbytes, pattern, mask, 0, 0, 0)
    cap = nbytes // frame_nbytes + 16
    offs = torch.empty(cap, dtype=torch.int64, device=dbuf.device)
    count = torch.zeros(1, dtype=torch.int64, device=dbuf.device)
    check(lib.bb_vdif_locate(_ptr(dbuf), nbytes, C.byref(p), _ptr(offs), cap,
                             _ptr(count), _stream(dbuf)), 'bb_vdif_locate')
    n = min(int(count.item()), cap)
    return torch.sort(offs[:n]).values


def vdif_scan_at(dbuf, nbytes, offsets, frame_nbytes, header_nbytes, pattern,
                 mask, ref_seconds, ref_frame_nr, frame_rate):
    p = _vdif_params(frame_nbytes, header_nbytes, pattern, mask, ref_seconds,
                     ref_frame_nr, frame_rate)
    n = offsets.numel()
    recs = torch.empty((n, 4), dtype=torch.int32, device=dbuf.device)
    check(lib.bb_vdif_scan_at(_ptr(dbuf), nbytes, C.byref(p), _ptr(offsets), n,
                              _ptr(recs), _stream(dbuf)), 'bb_vdif_scan_at')
    return recs


def mark5b_scan(dbuf, nframes, ref_seconds, ref_frame_nr, frame_rate,
                first_offset=0):
    p = _lib.Mark5BScanParams()
    p.first_offset = first_offset
    p.ref_seconds = ref_seconds
    p.ref_frame_nr = ref_frame_nr
    p.frame_rate = frame_rate
    recs = torch.empty((nframes, 4), dtype=torch.int32, device=dbuf.device)
    check(lib.bb_mark5b_scan(_ptr(dbuf), dbuf.numel(), C.byref(p), _ptr(recs),
                             nframes, _stream(dbuf)), 'bb_mark5b_scan')
    return recs


def _collect_offsets(launch, dbuf, cap, where):
    offs = torch.empty(cap, dtype=torch.int64, device=dbuf.device)
    count = torch.zeros(1, dtype=torch.int64, device=dbuf.device)
    check(launch(_ptr(offs), cap, _ptr(count)), where)
    n = min(int(count.item()), cap)
    return torch.sort(offs[:n]).values


def mark5b_locate(dbuf, nbytes, w1_pattern=0, w1_mask=0):
    """Byte-granular Mark 5B frame search -> sorted int64 device offsets.  With a mask, word 1 of
    the headers must agree with the pattern under it (the stream's user word)."""
    return _collect_offsets(
        lambda offs, cap, count: lib.bb_mark5b_locate_stream(_ptr(dbuf), nbytes, int(w1_pattern) & 0xffffffff,
                                                             int(w1_mask) & 0xffffffff, offs, cap, count,
                                                             _stream(dbuf)),
        dbuf, nbytes // 10016 + 16, 'bb_mark5b_locate_stream')


def mark5b_scan_at(dbuf, nbytes, offsets, ref_seconds, ref_frame_nr, frame_rate):
    p = _lib.Mark5BScanParams()
    p.first_offset, p.ref_seconds = 0, ref_seconds
    p.ref_frame_nr, p.frame_rate = ref_frame_nr, frame_rate
    n = offsets.numel()
    recs = torch.empty((n, 4), dtype=torch.int32, device=dbuf.device)
    check(lib.bb_mark5b_scan_at(_ptr(dbuf), nbytes, C.byref(p), _ptr(offsets), n,
                                _ptr(recs), _stream(dbuf)), 'bb_mark5b_scan_at')
    return recs


def mark4_locate(dbuf, nbytes, ntrack):
    """Byte-granular Mark 4 frame search -> sorted int64 device offsets."""
    return _collect_offsets(
        lambda offs, cap, count: lib.bb_mark4_locate(_ptr(dbuf), nbytes, ntrack, offs, cap, count,
                                                     _stream(dbuf)),
        dbuf, nbytes // (ntrack * 2500) + 16, 'bb_mark4_locate')


def mark4_header_crc(dbuf, nframes, ntrack, first_offset=0, offsets=None):
    """Per frame the set of tracks (bit mask, int64 tensor) whose 160 header
    bits fail the CRC-12 along the track (bb_mark4_header_crc; the reference's
    ``crc12.check``, mark4/header.py:34-44).  Never alters decoded samples."""
    bad = torch.empty(nframes, dtype=torch.int64, device=dbuf.device)
    check(lib.bb_mark4_header_crc(_ptr(dbuf), dbuf.numel(), ntrack, _ptr(offsets), first_offset,
                                  nframes, _ptr(bad), _stream(dbuf)), 'bb_mark4_header_crc')
    return bad


def mark4_scan_at(dbuf, nbytes, offsets, ntrack, ref_year, ref_qms, frame_qms):
    p = _lib.Mark4ScanParams()
    p.first_offset, p.ntrack, p.ref_year = 0, ntrack, ref_year
    p.ref_qms, p.frame_qms = ref_qms, frame_qms
    n = offsets.numel()
    recs = torch.empty((n, 4), dtype=torch.int32, device=dbuf.device)
    check(lib.bb_mark4_scan_at(_ptr(dbuf), nbytes, C.byref(p), _ptr(offsets), n,
                               _ptr(recs), _stream(dbuf)), 'bb_mark4_scan_at')
    return recs


def verify_records(recs, nrecs, first_index, recs_per_index, nstrict, nbad):
    """Add to the device counter `nbad` (int32[1]) the number of records that
    fail verification (see bb_verify_records)."""
    check(lib.bb_verify_records(_ptr(recs), nrecs, first_index, recs_per_index, nstrict,
                                _ptr(nbad), _stream(recs)), 'bb_verify_records')


def fetch_counter(counter, host, event, side_stream):
    """``host[0] = counter[0]`` fetched on `side_stream` once `event` has
    happened (bb_fetch_counter); returns when the value is there.  `host`: a
    pinned int32 tensor; `event`: torch.cuda.Event that has been recorded."""
    check(lib.bb_fetch_counter(_ptr(counter), C.c_void_p(host.data_ptr()),
                               C.c_void_p(event.cuda_event) if event is not None else C.c_void_p(0),
                               C.c_void_p(side_stream.cuda_stream)), 'bb_fetch_counter')


def recs_fields(recs):
    """Split scan records (device int32 (n,4)) into named host arrays."""
    r = recs.cpu().numpy()
    off = r[:, :2].copy().view(np.int64)[:, 0]
    tidx = r[:, 2]
    thread = (r[:, 3] & 0xffff).astype(np.int16)
    flags = ((r[:, 3] >> 16) & 0xffff).astype(np.uint16)
    return dict(payload_offset=off, time_index=tidx, thread_id=thread, flags=flags)


def thread_slot_map(thread_ids, device):
    m = np.full(1024, -1, dtype=np.int16)
    for s, t in enumerate(thread_ids):
        m[int(t)] = s
    return torch.from_numpy(m).to(device)


def build_index(recs, nframes_out, nslot=1, thread_slot=None):
    src = torch.empty(nframes_out * nslot, dtype=torch.int64, device=recs.device)
    check(lib.bb_build_index(_ptr(recs), recs.shape[0], _ptr(thread_slot), nslot,
                             _ptr(src), nframes_out, _stream(recs)), 'bb_build_index')
    return src


def decode_frames(dbuf, nframes, payload_nbytes, coder, bps, chunk=1, nslot=1,
                  src=None, src0=0, src_stride=0, complex_data=False,
                  fill_value=0., out=None, within=None):
    """Decode `nframes` frame(set)s to a flat float32 device tensor.  `within`
    (int32 device tensor of positions inside a thread sample's `chunk`
    floats) selects channels in the kernel: only those positions are
    written (bb_decode_frames_select)."""
    p = _lib.DecodeParams()
    p.coder = coder
    p.bps = bps
    p.chunk = chunk
    p.nslot = nslot
    p.payload_nbytes = payload_nbytes
    p.src0 = src0
    p.src_stride = src_stride
    p.complex_data = int(bool(complex_data))
    fv = complex(fill_value)
    p.fill_re = fv.real
    p.fill_im = fv.imag
    nelem = nframes * nslot * (payload_nbytes * 8 // bps) if bps in (1, 2, 4, 8) else 0
    if within is not None:
        if src is None:                 # the selecting kernel works through an index
            src = src0 + torch.arange(nframes * nslot, dtype=torch.int64, device=dbuf.device) * src_stride
        nsel = within.numel()
        nelem = nelem // chunk * nsel
        if out is None:
            out = _empty_output(nelem, dbuf.device)
        check(lib.bb_decode_frames_select(_ptr(dbuf), dbuf.numel(), _ptr(src), nframes, C.byref(p),
                                          _ptr(within), nsel, _ptr(out), out.numel(), _stream(dbuf)),
              'bb_decode_frames_select')
        return out
    tgt = _Target(out, nelem, dbuf.device)
    check(lib.bb_decode_frames(_ptr(dbuf), dbuf.numel(), _ptr(src), nframes,
                               C.byref(p), _ptr(tgt.use), tgt.use.numel(), _stream(dbuf)),
          'bb_decode_frames')
    return tgt.done()


def copy_frames(dbuf, nframes, nbytes_per_frame, src0=0, src_stride=0, out=None):
    """`nframes` runs of `nbytes_per_frame` bytes of `dbuf` (uint8) at ``src0 +
    f * src_stride`` -> one contiguous float32 device tensor (bb_copy_frames):
    float32 samples passed through as they are.  EXTENSION -- the reference
    has no NBIT 32 decoder (dada/payload.py:40-41)."""
    nelem = nframes * nbytes_per_frame // 4
    tgt = _Target(out, nelem, dbuf.device)
    check(lib.bb_copy_frames(_ptr(dbuf), dbuf.numel(), nframes, nbytes_per_frame, src0, src_stride,
                             _ptr(tgt.use), tgt.use.numel() * 4, _stream(dbuf)), 'bb_copy_frames')
    return tgt.done()


TOUCH_MIN_BYTES = 16 << 20
def _touch_max_bytes():
    try:                                # (as the library reads it: csrc/bbdecode.hip touch_mib_default)
        mib = int(os.environ.get('BB_TOUCH_MIB', '') or 256)
    except ValueError:
        mib = 256
    return (256 if mib < 0 else mib) << 20


TOUCH_MAX_BYTES = _touch_max_bytes()


def touch(dbuf, lo, nbytes):
    """Queue a pass that reads bytes [lo, lo + nbytes) of the uint8 device tensor `dbuf` once
    and keeps nothing (bb_touch): a decode of those bytes queued right behind it finds them in
    the device's memory-side cache.  Does nothing outside 16 MiB .. BB_TOUCH_MIB (256)."""
    if TOUCH_MIN_BYTES <= nbytes <= TOUCH_MAX_BYTES:
        check(lib.bb_touch(C.c_void_p(dbuf.data_ptr() + int(lo)), int(nbytes), _stream(dbuf)), 'bb_touch')


def select_supported(bps, chunk, nslot, nwithin, payload_nbytes=None):
    """Would `decode_frames(..., within=...)` take this geometry?  Asks the
    library's own argument / geometry checks (bb_decode_frames_select_check;
    no device needed), so that readers fold a channel subset into the decode
    only when the kernel accepts it and keep the reference's
    decode-then-index order otherwise.  `payload_nbytes` None: a payload long
    enough not to limit the work items (requests of varying length)."""
    p = _lib.DecodeParams()
    p.coder = _lib.CODER_VDIF               # (any coder that has this sample width: the limits do not depend on it)
    p.bps = bps
    p.chunk = chunk
    p.nslot = nslot
    if payload_nbytes is None:
        payload_nbytes = max(1 << 20, chunk * bps // 8 * 4) if bps in (1, 2, 4, 8) else 0
    p.payload_nbytes = payload_nbytes
    return lib.bb_decode_frames_select_check(C.byref(p), int(nwithin)) == _lib.BB_OK


def mark4_scan(dbuf, nframes, ntrack, ref_year, ref_qms, frame_qms,
               first_offset=0):
    p = _lib.Mark4ScanParams()
    p.first_offset = first_offset
    p.ntrack = ntrack
    p.ref_year = ref_year
    p.ref_qms = ref_qms
    p.frame_qms = frame_qms
    recs = torch.empty((nframes, 4), dtype=torch.int32, device=dbuf.device)
    check(lib.bb_mark4_scan(_ptr(dbuf), dbuf.numel(), C.byref(p), _ptr(recs),
                            nframes, _stream(dbuf)), 'bb_mark4_scan')
    return recs


def mark4_select_maps(sign_bit, mag_bit, nchan, channels):
    """Bit maps of a Mark 4 mode restricted to `channels`: output ``fo * nchan
    + c`` of a stream word is sample `fo`, channel `c`; the result lists, for
    every sample of the word, the kept channels in the order given."""
    fanout = len(sign_bit) // nchan
    keep = [fo * nchan + int(c) for fo in range(fanout) for c in channels]
    return [sign_bit[j] for j in keep], [mag_bit[j] for j in keep]


def decode_mark4(dbuf, nframes, ntrack, nwords, sign_bit, mag_bit, fill_words=0,
                 src=None, src0=0, src_stride=0, fill_value=0., out=None, select=False):
    """Track-demultiplex `nframes` units of `nwords` stream words each ->
    flat float32 device tensor of nframes * nwords * ntrack/2 values.  With
    `select` the maps may be shorter (the outputs of the channels a reader's
    subset keeps, `mark4_select_maps`): every word then gives ``len(sign_bit)``
    values (bb_decode_mark4_select)."""
    p = _lib.Mark4DecodeParams()
    p.ntrack = ntrack
    p.nwords = nwords
    p.fill_words = fill_words
    p.src0 = src0
    p.src_stride = src_stride
    for j, (s, m) in enumerate(zip(sign_bit, mag_bit)):
        p.sign_bit[j] = s
        p.mag_bit[j] = m
    p.fill = float(fill_value)
    if select:
        nout = len(sign_bit)
        nelem = nframes * nwords * nout
        if out is None:
            out = _empty_output(nelem, dbuf.device)
        check(lib.bb_decode_mark4_select(_ptr(dbuf), dbuf.numel(), _ptr(src), nframes, C.byref(p),
                                         nout, _ptr(out), out.numel(), _stream(dbuf)),
              'bb_decode_mark4_select')
        return out
    tgt = _Target(out, nframes * nwords * (ntrack // 2), dbuf.device)
    check(lib.bb_decode_mark4(_ptr(dbuf), dbuf.numel(), _ptr(src), nframes,
                              C.byref(p), _ptr(tgt.use), tgt.use.numel(), _stream(dbuf)),
          'bb_decode_mark4')
    return tgt.done()


def decode_i8_tiled(dbuf, nframes, layout, npol, nchan, ntime, t_lo, t_hi,
                    src=None, src0=0, src_stride=0, fill_value=0., out=None, nchan_stored=0,
                    npol_stored=0, pol_first=0, chan_map=None):
    """int8 (re, im) -> complex64 with the (time, pol, chan) permutation of
    `layout`; rows [t_lo, t_hi) of every frame -> flat float32 tensor.  With
    `nchan_stored` the payload holds that many channels and the `nchan`
    starting at the payload offset are decoded (`tiled_channel_skip`).  A
    SELECTION -- `chan_map` (int32 device tensor of `nchan` stored-channel
    numbers) and / or `npol` of `npol_stored` polarisations from `pol_first` on
    -- is decoded from payload offsets that point at the payload start;
    KeyError when the library's fast form does not take the geometry."""
    p = _lib.TiledParams()
    p.layout = layout
    p.npol = npol
    p.nchan = nchan
    p.nchan_stored = nchan_stored
    p.npol_stored = npol_stored
    p.pol_first = pol_first
    p.d_chan_map = chan_map.data_ptr() if chan_map is not None else None
    p.ntime = ntime
    p.t_lo = t_lo
    p.t_hi = t_hi
    p.src0 = src0
    p.src_stride = src_stride
    fv = complex(fill_value)
    p.fill_re = fv.real
    p.fill_im = fv.imag
    nelem = nframes * (t_hi - t_lo) * npol * nchan * 2
    tgt = _Target(out, nelem, dbuf.device)
    check(lib.bb_decode_i8_tiled(_ptr(dbuf), dbuf.numel(), _ptr(src), nframes,
                                 C.byref(p), _ptr(tgt.use), tgt.use.numel(), _stream(dbuf)),
          'bb_decode_i8_tiled')
    return tgt.done()


def tiled_channel_skip(layout, npol, ntime, c_lo):
    """Bytes from the start of a payload to its channel `c_lo`, for
    `decode_i8_tiled(..., nchan_stored=...)`."""
    if layout == _lib.LAYOUT_GUPPI_CF:
        return c_lo * ntime * npol * 2          # every channel is one run of all times
    if layout == _lib.LAYOUT_MKBF:
        return c_lo * 256 * 2                   # 256 times of a channel inside a (heap, pol) block
    return c_lo * npol * 2                      # channels of one time lie side by side


def as_device_samples(data):
    """Samples to encode -> float32 / complex64 tensor on the GPU (NumPy
    arrays and host tensors are uploaded)."""
    require_gpu()
    if not isinstance(data, torch.Tensor):
        from .staging import upload_array
        data = upload_array(data)
    want = torch.complex64 if data.is_complex() else torch.float32
    return data.to(device='cuda', dtype=want)


def encode_flat(values, coder, bps):
    """float32 (or complex64) device tensor -> packed uint8 device tensor."""
    require_gpu()
    if values.is_complex():
        values = torch.view_as_real(values)
    values = values.to(torch.float32).contiguous().reshape(-1)
    nbytes = values.numel() * bps // 8 if bps in (1, 2, 4, 8) else 0
    # the kernel packs quads of samples into whole bytes: a shorter tail (item
    # assignment of a few samples) is padded here and cut off the output
    unit = max(4, 8 // bps) if bps in (1, 2, 4, 8) else 4
    short = -values.numel() % unit
    if short and (values.numel() * bps) % 8 == 0:       # whole bytes only
        values = torch.nn.functional.pad(values, (0, short))
    out = torch.empty(values.numel() * bps // 8 if bps in (1, 2, 4, 8) else 0,
                      dtype=torch.uint8, device=values.device)
    check(lib.bb_encode_flat(_ptr(values), values.numel(), coder, bps, _ptr(out),
                             out.numel(), _stream(values)), 'bb_encode_flat')
    return out[:nbytes]


def encode_mark4(values, ntrack, sign_bit, mag_bit):
    """(nsample, nchan) float32 device tensor -> stream words as uint8."""
    require_gpu()
    values = values.to(torch.float32).contiguous().reshape(-1)
    opw = ntrack // 2
    nwords = values.numel() // opw
    sb = (C.c_uint8 * 32)(*sign_bit)
    mb = (C.c_uint8 * 32)(*mag_bit)
    out = torch.empty(nwords * (ntrack // 8), dtype=torch.uint8, device=values.device)
    check(lib.bb_encode_mark4(_ptr(values), nwords, ntrack, sb, mb, _ptr(out),
                              out.numel(), _stream(values)), 'bb_encode_mark4')
    return out


def tune(knob, value):
    check(lib.bb_tune(knob, value), 'bb_tune')
