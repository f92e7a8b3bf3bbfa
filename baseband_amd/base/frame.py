"""Frame = header + payload, with the validity -> fill_value rule.

Mirror of the reference's ``FrameBase`` (base/frame.py:14-241): a frame acts
as a dict of header keys, indexes/slices like its payload, and returns
``fill_value`` everywhere when it is not valid (base/frame.py:191-199).
Decoded data are device tensors.
"""
import numpy as np
import torch


class FrameBase:
    _header_class = None
    _payload_class = None
    _fill_value = 0.
    _valid = True

    def __init__(self, header, payload, valid=None, verify=True):
        self.header = header
        self.payload = payload
        if valid is not None:
            self.valid = valid
        if verify:
            self.verify()

    def verify(self):
        assert isinstance(self.header, self._header_class)
        assert isinstance(self.payload, self._payload_class)
        payload_nbytes = getattr(self.header, 'payload_nbytes', None)
        if payload_nbytes is not None:
            assert self.payload.nbytes == payload_nbytes

    @property
    def valid(self):
        return self._valid

    @valid.setter
    def valid(self, valid):
        self._valid = bool(valid)

    @classmethod
    def fromfile(cls, fh, memmap=None, valid=None, verify=True, **kwargs):
        header = cls._header_class.fromfile(fh, verify=verify)
        payload = cls._payload_class.fromfile(fh, header=header, memmap=memmap,
                                              **kwargs)
        return cls(header, payload, valid=valid, verify=verify)

    def tofile(self, fh):
        self.header.tofile(fh)
        self.payload.tofile(fh)

    @property
    def sample_shape(self):
        return self.payload.sample_shape

    def __len__(self):
        return len(self.payload)

    @property
    def shape(self):
        return (len(self),) + tuple(self.sample_shape)

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
    def dtype(self):
        return self.payload.dtype

    @property
    def nbytes(self):
        return self.header.nbytes + self.payload.nbytes

    @property
    def fill_value(self):
        return self._fill_value

    @fill_value.setter
    def fill_value(self, fill_value):
        self._fill_value = fill_value

    def _fill(self, shape):
        tdtype = torch.complex64 if self.dtype.kind == 'c' else torch.float32
        return torch.full(tuple(shape), self.fill_value, dtype=tdtype,
                          device='cuda')

    def __getitem__(self, item=()):
        if isinstance(item, str):
            return self.header.__getitem__(item)
        if self.valid:
            return self.payload[item]
        # shape of the item without decoding anything
        probe = np.empty(self.shape, dtype=bool)[item]
        return self._fill(probe.shape)

    data = property(__getitem__, doc="Full decoded frame (device tensor).")

    def keys(self):
        return self.header.keys()

    def __contains__(self, key):
        return key in self.header.keys()

    def __getattr__(self, attr):
        if attr in ('header', 'payload'):
            raise AttributeError(attr)
        try:
            return getattr(self.header, attr)
        except AttributeError:
            raise AttributeError("{} object has no attribute {}"
                                 .format(type(self).__name__, attr))

    def __eq__(self, other):
        return (type(self) is type(other)
                and self.valid == other.valid
                and self.header == other.header
                and self.payload == other.payload)
