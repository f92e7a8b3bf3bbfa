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
    def fromdata(cls, data, header=None, verify=True, **kwargs):
        if header is None:
            header = VDIFHeader.fromvalues(verify=verify, **kwargs)
        payload = VDIFPayload.fromdata(data, header=header)
        return cls(header, payload, verify=verify)


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
        block = b''
        nwant = 9                       # frames per block, grown geometrically
        while True:
            ask = nwant * fn + 32 - len(block)
            more = fh.read(ask)
            block += more
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
                    raise
                end += hn
        else:
            # no further header could be read
            if not complete:
                if nset < nhead:        # it is there but damaged: raises
                    VDIFHeader(np.frombuffer(block, '<u4', 8, nset * fn), edv, verify=False).verify()
                raise EOFError
            # the pointer ends where that attempt left it: behind a damaged
            # header, or at the end of a tail too short to hold one
            if nset < nhead:
                end += hn
            elif start + len(block) - end < 32:
                end = start + len(block)
        fh.seek(end)
        if wanted is not None and len(frames) < len(wanted):
            raise OSError("could not find all requested frames.")
        order = sorted(frames) if thread_ids is None else list(thread_ids)
        return cls([frames[tid] for tid in order], header0)

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
            headers = VDIFHeader.fromvalues(
                complex_data=data.is_complex(), verify=verify, **kwargs)
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
        return any(f.valid for f in self.frames)

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

    def __getattr__(self, attr):
        if attr in ('frames', 'header0'):
            raise AttributeError(attr)
        return getattr(self.header0, attr)

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
