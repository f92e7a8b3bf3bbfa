"""GSB payloads (gsb/payload.py:56-150): 4-bit signed nibbles (rawdump) or
int8 pairs (phased) decoded on the GPU by ``bb_decode_frames`` with the INT
coder.  ``fromfile`` accepts the nested tuple of filehandles of phased data
and interleaves the parts exactly like gsb/payload.py:88-131."""
from collections import namedtuple

import numpy as np

from .. import _lib
from ..base.payload import PayloadBase

__all__ = ['GSBPayload']


class GSBPayload(PayloadBase):
    _dtype_word = np.dtype('i1')
    _coder_id = _lib.CODER_INT
    _sample_shape_maker_1thread = namedtuple('SampleShape', 'nchan')
    _sample_shape_maker_nthread = namedtuple('SampleShape', 'nthread, nchan')

    @classmethod
    def _sample_shape_maker(cls, *args):
        if len(args) == 1:
            return cls._sample_shape_maker_1thread(*args)
        return cls._sample_shape_maker_nthread(*args)

    def _decode(self, byte_start, byte_stop):
        if self.bps not in (4, 8):
            raise KeyError(self.bps)
        # byte ranges of int8 words need not be dword aligned
        lo = byte_start - byte_start % 4
        hi = -(-byte_stop // 4) * 4
        per = 8 // self.bps
        import torch
        from .. import kernels
        if byte_stop == byte_start:             # (an empty slice)
            return torch.empty(0, dtype=torch.float32, device='cuda')
        dbuf = self._device_words()
        if hi > dbuf.numel():
            dbuf = torch.nn.functional.pad(dbuf, (0, hi - dbuf.numel()))
        flat = kernels.decode_frames(dbuf, 1, hi - lo, _lib.CODER_INT, self.bps,
                                     src0=lo)
        return flat[(byte_start - lo) * per:(byte_stop - lo) * per]

    @classmethod
    def fromfile(cls, fh, *, payload_nbytes=1 << 22, sample_shape=(1,), bps=4,
                 complex_data=False):
        if hasattr(fh, 'read'):
            return super().fromfile(fh, payload_nbytes=payload_nbytes,
                                    sample_shape=sample_shape, bps=bps,
                                    complex_data=complex_data)
        parts = [[np.frombuffer(fh1.read(payload_nbytes), dtype=cls._dtype_word)
                  for fh1 in fh_set] for fh_set in fh]
        if any(len(p) < payload_nbytes for ps in parts for p in ps):
            raise EOFError("could not read full payload.")
        bpfs = bps * (2 if complex_data else 1) * int(np.prod(sample_shape[1:]))
        sample_nbytes, extra = divmod(bpfs, 8)
        assert extra == 0, 'Full samples do not fit in integer number of bytes'
        nfile = len(parts[0])
        words = np.empty((nfile, payload_nbytes // sample_nbytes,
                          sample_shape[0], sample_nbytes), dtype=cls._dtype_word)
        for p, pset in enumerate(parts):
            for f, part in enumerate(pset):
                words[f, :, p, :] = part.reshape(-1, sample_nbytes)
        return cls(words.ravel(), sample_shape=sample_shape, bps=bps,
                   complex_data=complex_data)

    def tofile(self, fh):
        """Write the words to one raw file, or split a phased payload over
        ``fh[pol][part]``: parts are consecutive in time, polarisations
        interleaved per sample (gsb/payload.py:133-144)."""
        if hasattr(fh, 'write'):
            return fh.write(self.words.tobytes())
        npol = len(fh)
        assert npol == self.sample_shape[0]
        words = self.words.reshape(len(fh[0]), -1, npol, self._bpfs // npol // 8)
        for fh_set, pol in zip(fh, words.transpose(2, 0, 1, 3)):
            for fh1, part in zip(fh_set, pol):
                fh1.write(np.ascontiguousarray(part).tobytes())
