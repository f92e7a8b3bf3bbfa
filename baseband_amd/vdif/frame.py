"""VDIF frames and frame sets.

Mirrors ``VDIFFrame`` (vdif/frame.py:21-128) and ``VDIFFrameSet``
(vdif/frame.py:131-512).  A frame set gathers the frames of all threads of
one time step; its ``data`` is ``(nsample, nthread, nchan)``.  Instead of
decoding each thread and copying it into a strided column
(vdif/frame.py:402-434), the set uploads the payloads once and runs a single
``bb_decode_frames`` launch with ``nslot = nthread`` that writes the
interleaved layout directly; invalid frames are handed over as source -1 and
come back as ``fill_value``.
"""
import numpy as np
import torch

from ..staging import to_numpy

from .. import kernels
from ..base.frame import FrameBase
from .header import VDIFHeader
from .payload import VDIFPayload

__all__ = ['VDIFFrame', 'VDIFFrameSet']


class VDIFFrame(FrameBase):
    _header_class = VDIFHeader
    _payload_class = VDIFPayload

    def __init__(self, header, payload, valid=None, verify=True):
        self.header = header
        self.payload = payload
        if valid is not None:
            self.valid = valid
        if verify:
            self.verify()

    @property
    def valid(self):
        """Valid unless the header's invalid_data bit is set
        (vdif/frame.py:79-90)."""
        return not self.header['invalid_data']

    @valid.setter
    def valid(self, valid):
        if bool(valid) == self.header['invalid_data']:
            if not self.header.mutable:
                self.header = self.header.copy()
            self.header['invalid_data'] = not valid

    def verify(self):
        super().verify()
        assert self.header.payload_nbytes == self.payload.nbytes
        assert self.payload.sample_shape == (self.header.nchan,)

    @classmethod
    def fromfile(cls, fh, edv=None, verify=True):
        header = VDIFHeader.fromfile(fh, edv, verify)
        payload = VDIFPayload.fromfile(fh, header=header)
        return cls(header, payload, verify=verify)

    @classmethod
    def fromdata(cls, data, header=None, *, valid=None, verify=True, **kwargs):
        # (``valid=False`` sets the header's invalid_data bit, as the reference's
        # FrameBase.fromdata does through the frame's ``valid``: base/frame.py:97-131)
        if header is None:
            header = VDIFHeader.fromvalues(verify=verify, **kwargs)
        payload = VDIFPayload.fromdata(data, header=header)
        return cls(header, payload, valid=valid, verify=verify)


def _from_mark5b_frame(cls, mark5b_frame, verify=True, **kwargs):
    """VDIF frame (EDV 0xab) around a Mark 5B frame: same payload words, the
    header converted (vdif/frame.py:104-128)."""
    m5_payload = mark5b_frame.payload
    header = VDIFHeader.from_mark5b_header(mark5b_frame.header, bps=m5_payload.bps,
                                           nchan=m5_payload.sample_shape[0],
                                           invalid_data=not mark5b_frame.valid, **kwargs)
    payload = VDIFPayload(m5_payload.words, header)
    return cls(header, payload, verify=verify)


VDIFFrame.from_mark5b_frame = classmethod(_from_mark5b_frame)


class VDIFFrameSet:
    def __init__(self, frames, header0=None):
        self.frames = frames
        self.header0 = frames[0].header if header0 is None else header0
        self._fill_value = 0.

    @classmethod
    def fromfile(cls, fh, thread_ids=None, edv=None, verify=True):
        """The frames of one time step, from the current file position; the
        pointer is left at the first frame that is not part of the set.

        Same results as the reference's frame-by-frame walk
        (vdif/frame.py:176-243): a set ends before the first header with
        another frame number or a thread id already seen, or where no further
        header can be read; with `thread_ids` given only those threads are
        kept and all of them must be there.  Done here on a TABLE: the set's
        bytes are read in one block (grown until the end of the set is inside
        it), the headers are looked at as a strided uint32 array -- frame
        number and thread id of every frame in two vector operations -- and
        the payloads are views into the block (what the stream path does per
        window with ``bb_vdif_scan`` / ``bb_build_index``, on the host because
        a frame set's words live on the host)."""
        start = fh.tell()
        header0 = VDIFHeader.fromfile(fh, edv, verify)
        edv = header0.edv
        fn, hn = header0.frame_nbytes, header0.nbytes
        wanted = None if thread_ids is None else set(thread_ids)
        fh.seek(start)
        # frames per block: the set that is asked for plus the header behind it
        # (a single-thread file: two frames, not nine), grown geometrically;
        # pieces are joined once per growth.  Every frame of a set
        # is taken to have header0's size, as the reference's readers do
        # (vdif/base.py: fixed `frame_nbytes` per stream).
        pieces, have = [], 0
        nwant = 2 if wanted is None else len(wanted) + 1
        while True:
            ask = nwant * fn + 32 - have
            more = fh.read(ask)
            pieces.append(more)
            have += len(more)
            block = pieces[0] if len(pieces) == 1 else b''.join(pieces)
            pieces = [block]
            raw = np.frombuffer(block, np.uint8)
            # headers that can be parsed: 32 bytes are read even for a legacy one
            nhead = (len(raw) - 32) // fn + 1
            table = np.lib.stride_tricks.as_strided(
                raw[:(nhead - 1) * fn + 32].view('<u4'), shape=(nhead, 4), strides=(fn, 4),
                writeable=False)
            frame_nr = table[:, 1] & 0xffffff
            thread = (table[:, 3] >> 16) & 0x3ff
            # only threads that are kept can end a set by showing up again
            kept = np.ones(nhead, bool) if wanted is None else np.isin(thread, sorted(wanted))
            again = kept.copy()
            again[np.nonzero(kept)[0][np.unique(thread[kept], return_index=True)[1]]] = False
            breaks = np.nonzero((frame_nr != frame_nr[0]) | again)[0]
            if len(breaks) or len(more) < ask:
                break
            nwant *= 4
        nset = int(breaks[0]) if len(breaks) else nhead      # frames whose header is in the set
        # the table takes every frame to have header0's size.  Where a header in
        # reach says otherwise (garbage behind a damaged frame, mixed legacy
        # headers), walk frame by frame, every header's own length honoured
        look = min(nset + 1, nhead)
        if (np.any((table[:look, 2] & 0xffffff) * 8 != fn)
                or np.any(((table[:look, 0] >> 30) & 1) != (table[0, 0] >> 30) & 1)):
            fh.seek(start + hn)
            frames = cls._walk(fh, header0, thread_ids, edv, verify)
            if wanted is not None and len(frames) < len(wanted):
                raise OSError("could not find all requested frames.")
            order = sorted(frames) if thread_ids is None else list(thread_ids)
            return cls([frames[tid] for tid in order], header0)
        eof = start + len(block)
        frames = {}
        for k in range(nset):
            if k == 0:
                header = header0
            else:
                header = VDIFHeader(np.frombuffer(block, '<u4', 8, k * fn), edv, verify=False)
                try:
                    if verify:
                        header.verify()
                except AssertionError:
                    nset = k            # a damaged header ends the set like the end of the file
                    break
            if kept[k]:
                body = raw[k * fn + hn:(k + 1) * fn]
                if len(body) < fn - hn:
                    raise EOFError("could not read full payload.")
                frames[int(thread[k])] = VDIFFrame(header, VDIFPayload(body.view('<u4'), header),
                                                   verify=False)
        complete = wanted is None or len(frames) == len(wanted)
        end = start + nset * fn
        if len(breaks) and nset == int(breaks[0]):
            # ended by a header of the next set; the reference has parsed and
            # verified that one too before looking at its frame number
            try:
                if verify:
                    VDIFHeader(np.frombuffer(block, '<u4', 8, nset * fn), edv, verify=False).verify()
            except AssertionError:
                if not complete:
                    fh.seek(cls._behind_header(start, block, nset * fn))
                    raise
                end = cls._behind_header(start, block, nset * fn)
        else:
            # no further header could be read.  The pointer ends where that
            # attempt left it: behind a damaged header; else at the end of a
            # tail too short to hold one -- or BEYOND the end of the file, when
            # the last frame was one that is skipped, not read, and is cut short
            if nset < nhead:
                behind = cls._behind_header(start, block, nset * fn)
            else:
                behind = max(end, eof)
            if not complete:
                fh.seek(behind)
                if nset < nhead:        # it is there but damaged: raises
                    VDIFHeader(np.frombuffer(block, '<u4', 8, nset * fn), edv, verify=False).verify()
                raise EOFError
            end = behind
        fh.seek(end)
        if wanted is not None and len(frames) < len(wanted):
            raise OSError("could not find all requested frames.")
        order = sorted(frames) if thread_ids is None else list(thread_ids)
        return cls([frames[tid] for tid in order], header0)

    @classmethod
    def _walk(cls, fh, header0, thread_ids, edv, verify):
        """The set frame by frame, from behind header0: a frame is read (wanted
        thread) or skipped by ITS header's length, then the next header is
        looked at, until one belongs to another set (the pointer goes back in
        front of it) or cannot be read (vdif/frame.py:203-234)."""
        frames, header, nr0 = {}, header0, header0['frame_nr']
        while header['frame_nr'] == nr0 and header['thread_id'] not in frames:
            tid = header['thread_id']
            if thread_ids is None or tid in thread_ids:
                frames[tid] = VDIFFrame(header, VDIFPayload.fromfile(fh, header=header), verify=False)
            else:
                fh.seek(header.payload_nbytes, 1)
            try:
                header = VDIFHeader.fromfile(fh, edv, verify)
            except (EOFError, AssertionError):
                if thread_ids is not None and len(frames) < len(thread_ids):
                    raise
                return frames
        fh.seek(-header.nbytes, 1)
        return frames

    @staticmethod
    def _behind_header(start, block, offset):
        """File position the reference is left at when the header at `offset`
        of the block fails verification: VDIFHeader.fromfile has read its four
        or -- unless the legacy bit is set -- eight words
        (vdif/header.py:158-186), then `verify` raised."""
        legacy = bool(np.frombuffer(block, '<u4', 1, offset)[0] & (1 << 30))
        return start + offset + (16 if legacy else 32)

    def tofile(self, fh):
        for frame in self.frames:
            frame.tofile(fh)

    @classmethod
    def fromdata(cls, data, headers=None, verify=True, **kwargs):
        """Encode (samples_per_frame, nthread, nchan) data as one frame per
        thread (vdif/frame.py:250-311)."""
        from .. import kernels
        data = kernels.as_device_samples(data)          # threads are encoded on the GPU
        nthread = data.shape[1]
        if headers is None:
            kwargs.setdefault('thread_id', 0)
            kwargs.setdefault('complex_data', data.is_complex())        # (``**header`` brings its own)
            headers = VDIFHeader.fromvalues(verify=verify, **kwargs)
        if isinstance(headers, VDIFHeader):
            header0 = headers
            headers = []
            for t in range(nthread):
                h = header0.copy()
                h['thread_id'] = header0['thread_id'] + t
                headers.append(h)
        frames = [VDIFFrame.fromdata(data[:, i], h, verify=verify)
                  for i, h in enumerate(headers)]
        return cls(frames)

    # -- properties
    @property
    def sample_shape(self):
        return (len(self.frames),) + tuple(self.frames[0].sample_shape)

    def __len__(self):
        return len(self.frames[0])

    @property
    def shape(self):
        return (len(self),) + self.sample_shape

    @property
    def dtype(self):
        return self.frames[0].dtype

    @property
    def nbytes(self):
        return len(self.frames) * self.frames[0].nbytes

    @property
    def size(self):
        size = 1
        for dim in self.shape:
            size *= dim
        return size

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def valid(self):
        """Whether the frames hold valid data: one bool when they agree, else one per
        frame (vdif/frame.py:329-339)."""
        valid = np.array([bool(f.valid) for f in self.frames])
        return bool(valid[0]) if len(np.unique(valid)) == 1 else valid

    @valid.setter
    def valid(self, valid):
        for f, v in zip(self.frames, np.broadcast_to(valid, (len(self.frames),))):
            f.valid = bool(v)

    @property
    def fill_value(self):
        return self._fill_value

    @fill_value.setter
    def fill_value(self, fill_value):
        self._fill_value = fill_value
        for f in self.frames:
            f.fill_value = fill_value

    def keys(self):
        return self.header0.keys()

    def __contains__(self, key):
        return key in self.header0

    def __eq__(self, other):
        return (type(self) is type(other) and len(self.frames) == len(other.frames)
                and self.header0 == other.header0
                and all(f1 == f2 for f1, f2 in zip(self.frames, other.frames)))

    __hash__ = None

    def __getattr__(self, attr):
        """Header properties are the set's: those every VDIF header has from the first
        header, the others (``sample_rate`` ...) from all frames -- one value when they
        agree, else one per frame (vdif/frame.py:498-505)."""
        if attr in ('frames', 'header0') or attr.startswith('__'):
            raise AttributeError(attr)
        header0 = self.header0
        if attr in type(header0)._properties:
            from .header import VDIFBaseHeader
            if attr in VDIFBaseHeader._properties:
                return getattr(header0, attr)
            values = [getattr(f.header, attr) for f in self.frames]
            if all(v == values[0] for v in values[1:]):
                return values[0]
            return np.array(values)
        raise AttributeError("{!r} object has no attribute {!r}".format(type(self).__name__, attr))

    def _decode_all(self):
        """All threads in one launch -> (nsample, nthread, nchan) tensor."""
        f0 = self.frames[0]
        pl = f0.payload
        nthread = len(self.frames)
        nbytes = pl.nbytes
        staged = np.empty(nthread * nbytes, dtype=np.uint8)
        src = np.empty(nthread, dtype=np.int64)
        for i, f in enumerate(self.frames):
            staged[i * nbytes:(i + 1) * nbytes] = f.payload.words.view(np.uint8)
            src[i] = i * nbytes if f.valid else -1
        dbuf = kernels.to_device_bytes(staged)
        dsrc = torch.from_numpy(src).to(dbuf.device)
        chunk = f0.header.nchan * (2 if pl.complex_data else 1)
        flat = kernels.decode_frames(
            dbuf, 1, nbytes, pl._coder_id, pl.bps, chunk=chunk, nslot=nthread,
            src=dsrc, complex_data=pl.complex_data, fill_value=self.fill_value)
        if pl.complex_data:
            flat = torch.view_as_complex(flat.view(-1, 2))
        nsample = len(pl)
        full = flat.reshape(-1, nthread, f0.header.nchan)
        return full[:nsample]

    def __getitem__(self, item=()):
        if isinstance(item, str):
            if item == 'thread_id':
                return np.array([f.header[item] for f in self.frames])
            if item != 'invalid_data':
                return self.header0[item]
            values = np.array([f.header[item] for f in self.frames])
            return values[0] if len(np.unique(values)) == 1 else values
        return self._decode_all()[item]

    def __setitem__(self, item, data):
        """Header keys for all frames, or samples in the (nsample, nthread,
        nchan) view of the set (vdif/frame.py:436-486): the touched threads'
        payloads are packed again by the GPU encoder."""
        if isinstance(item, str):
            if isinstance(data, (int, np.integer)):
                data = [int(data)] * len(self.frames)
                len_unique = 1
            elif (isinstance(data, (tuple, list))
                  and all(isinstance(d, (int, np.integer)) for d in data)):
                len_unique = len(set(data))
            else:
                raise ValueError("header items can only be set to integers.")
            if item == 'thread_id':
                if len_unique != len(self.frames):
                    raise ValueError("all thread ids should be unique.")
            elif (item != 'invalid_data' and len_unique > 1
                  and item in ('invalid_data', 'legacy_mode', 'seconds', '_1_30_2',
                               'ref_epoch', 'frame_nr', 'vdif_version', 'lg2_nchan',
                               'frame_length', 'complex_data', 'bits_per_sample',
                               'thread_id', 'station_id')):
                raise ValueError("base header keys should be identical.")
            for f, value in zip(self.frames, data):
                f.header[item] = value
            return
        full = self._decode_all().clone()
        if not isinstance(data, torch.Tensor):
            data = torch.from_numpy(np.ascontiguousarray(data))
        touched = torch.zeros(full.shape, dtype=torch.bool, device=full.device)
        touched[item] = True
        full[item] = data.to(device=full.device, dtype=full.dtype)
        for t in torch.nonzero(touched.any(0).any(-1)).flatten().tolist():
            self.frames[t].payload[:] = full[:, t]

    data = property(__getitem__, doc="Decoded frame set (device tensor).")

    def __array__(self, dtype=None, copy=None):
        host = to_numpy(self.data)
        return host if dtype in (None, host.dtype) else host.astype(dtype)
