"""GSB stream reader and ``open`` (gsb/base.py:146-387).

GSB data are headerless fixed-size blocks in one (rawdump) or several
(phased: polarisations x parts) raw files, with one line per block in a
separate timestamp file.  Frame k lives at ``k * payload_nbytes`` in every
raw file.  The phased layout -- parts consecutive in time, polarisations
interleaved per sample -- is exactly the multi-slot layout of
``bb_decode_frames``: output "frame" (k, part) with one slot per
polarisation.
"""
import io

import numpy as np
import torch

from .. import _lib, kernels
from ..base.base import GPUStreamReaderBase
from ..staging import host_image
from .header import GSBHeader
from .payload import GSBPayload
from .frame import GSBFrame

__all__ = ['GSBStreamReader', 'open']

DEFAULT_FRAME_RATE = 1e8 / 6 / 2 ** 22          # Hz (gsb/base.py:170)


class GSBStreamReader(GPUStreamReaderBase):
    def __init__(self, fh_ts, fh_raw, sample_rate=None, samples_per_frame=None,
                 payload_nbytes=None, nchan=None, bps=None, complex_data=None,
                 squeeze=True, subset=(), verify=True):
        self.fh_ts = fh_ts
        lines = [ln for ln in fh_ts.read().splitlines() if ln.strip()]
        lines = [ln.decode('ascii') if isinstance(ln, bytes) else ln for ln in lines]
        header0 = GSBHeader(lines[0].split())
        rawdump = header0.mode == 'rawdump'
        if isinstance(fh_raw, (tuple, list)):
            assert not rawdump
            for pair in fh_raw:
                assert isinstance(pair, (tuple, list))
                assert len(pair) == len(fh_raw[0])
        elif not rawdump:
            fh_raw = ((fh_raw,),)
        complex_data = (complex_data if complex_data is not None
                        else (False if rawdump else True))
        bps = bps if bps is not None else (4 if rawdump else 8)
        nchan = nchan if nchan is not None else (1 if rawdump else 512)
        bpfs = bps * nchan * (2 if complex_data else 1)
        nfiles = 1 if rawdump else len(fh_raw[0])
        if payload_nbytes is None:
            if samples_per_frame is None:
                if sample_rate is None:
                    payload_nbytes = 2 ** 22
                else:
                    payload_nbytes = int(round(sample_rate / DEFAULT_FRAME_RATE
                                               * bpfs / 8 / nfiles))
            else:
                payload_nbytes = samples_per_frame * bpfs // (8 * nfiles)
        if samples_per_frame is None:
            samples_per_frame = payload_nbytes * 8 // bpfs * nfiles
        elif samples_per_frame != payload_nbytes * nfiles * 8 / bpfs:
            raise ValueError('inconsistent samples_per_frame, bps, '
                             'complex_data, and payload_nbytes')
        if sample_rate is None:
            sample_rate = samples_per_frame * DEFAULT_FRAME_RATE
        shape = (nchan,) if rawdump else (len(fh_raw), nchan)
        super().__init__(
            fh_raw, header0, sample_rate=float(sample_rate),
            samples_per_frame=samples_per_frame, unsliced_shape=shape, bps=bps,
            complex_data=complex_data, squeeze=squeeze, subset=subset,
            fill_value=0., verify=verify)
        self._payload_nbytes = payload_nbytes
        self._rawdump = rawdump
        self._nfiles = nfiles
        # last usable timestamp line (gsb/base.py:314-347)
        last = GSBHeader(lines[-1].split(), verify=False)
        try:
            last.verify()
            assert len(' '.join(last.words)) >= len(' '.join(header0.words))
            last.time
        except Exception:
            last = GSBHeader(lines[-2].split())
        if rawdump:
            dt = (last.time - header0.time) / np.timedelta64(1, 'ns') * 1e-9
            nframes = int(round(dt * self.sample_rate / samples_per_frame)) + 1
        else:
            nframes = last['seq_nr'] - header0['seq_nr'] + 1
        self._nsample = nframes * samples_per_frame
        self._start_time = header0.time
        if rawdump:
            self._images = [[host_image(fh_raw)]]
        else:
            self._images = [[host_image(fh) for fh in pair] for pair in fh_raw]

    @property
    def payload_nbytes(self):
        return self._payload_nbytes

    def close(self):
        self._closed = True
        self.fh_ts.close()
        if self._rawdump:
            self.fh_raw.close()
        else:
            for pair in self.fh_raw:
                for fh in pair:
                    fh.close()

    def _read_sets(self, first, last):
        kernels.require_gpu()
        nsets = last - first
        pn = self._payload_nbytes
        npol = len(self._images)
        F = self._nfiles
        # stage [first, last) blocks of every raw file back to back
        staged = np.empty((npol, F, nsets * pn), dtype=np.uint8)
        for p in range(npol):
            for f in range(F):
                staged[p, f] = self._images[p][f][first * pn:last * pn]
        dbuf = kernels.to_device_bytes(staged.reshape(-1))
        nchan = self._unsliced_shape[-1]
        chunk = nchan * (2 if self.complex_data else 1)
        if self._rawdump:
            flat = kernels.decode_frames(dbuf, nsets, pn, _lib.CODER_INT, self.bps,
                                         chunk=chunk, src0=0, src_stride=pn)
        else:
            # output frame (k, part f), slot = polarisation p
            k = np.arange(nsets)[:, None, None]
            f = np.arange(F)[None, :, None]
            p = np.arange(npol)[None, None, :]
            src = ((p * F + f) * nsets + k) * pn
            dsrc = torch.from_numpy(np.ascontiguousarray(src.reshape(-1)).astype(np.int64)).to(dbuf.device)
            flat = kernels.decode_frames(dbuf, nsets * F, pn, _lib.CODER_INT,
                                         self.bps, chunk=chunk, nslot=npol,
                                         src=dsrc, complex_data=self.complex_data)
        if self.complex_data:
            flat = torch.view_as_complex(flat.view(-1, 2))
        return flat.reshape((nsets * self.samples_per_frame,)
                            + tuple(self._unsliced_shape))


def open(name, mode='rs', **kwargs):
    """Open a GSB timestamp file plus ``raw=`` data file(s) for stream
    reading (gsb/base.py:470-560)."""
    if mode != 'rs':
        raise ValueError("only stream reading mode 'rs' is supported "
                         "(got {!r}).".format(mode))
    raw = kwargs.pop('raw')
    fh_ts = name if hasattr(name, 'read') else io.open(name, 'r')
    if isinstance(raw, (tuple, list)):
        fh_raw = tuple(tuple(f if hasattr(f, 'read') else io.open(f, 'rb')
                             for f in (pair if isinstance(pair, (tuple, list)) else (pair,)))
                       for pair in raw)
    else:
        fh_raw = raw if hasattr(raw, 'read') else io.open(raw, 'rb')
    return GSBStreamReader(fh_ts, fh_raw, **kwargs)
