"""A frame couples one header with one payload.

Same call shapes as the reference's ``FrameBase`` (base/frame.py:14-241):
``fromfile``/``tofile``, dict-style access to header fields, array-style access
to decoded samples, ``valid`` and ``fill_value``.  The one rule that matters on
the decode path is the validity rule: an invalid frame reads as ``fill_value``
everywhere, without touching its payload (base/frame.py:191-199).  Samples come
back as device tensors.
"""
import numpy as np
import torch

from ..staging import to_numpy


def _delegate(owner, name, doc=None):
    """Read-only attribute forwarded to ``self.<owner>.<name>``."""
    return property(lambda self: getattr(getattr(self, owner), name), doc=doc)


class FrameBase:
    _header_class = None
    _payload_class = None
    _fill_value = 0.        # class defaults; instances override on assignment
    _valid = True

    def __init__(self, header, payload, valid=None, verify=True):
        self.header, self.payload = header, payload
        if valid is not None:
            self.valid = valid
        if verify:
            self.verify()

    def verify(self):
        """Header and payload have the right types and agree on the size."""
        assert isinstance(self.header, self._header_class)
        assert isinstance(self.payload, self._payload_class)
        expected = getattr(self.header, 'payload_nbytes', None)
        assert expected is None or expected == self.payload.nbytes

    # -- construction / serialisation
    @classmethod
    def fromfile(cls, fh, memmap=None, valid=None, verify=True, **kwargs):
        header = cls._header_class.fromfile(fh, verify=verify)
        payload = cls._payload_class.fromfile(fh, header=header, memmap=memmap,
                                              **kwargs)
        return cls(header, payload, valid=valid, verify=verify)

    def tofile(self, fh):
        for part in (self.header, self.payload):
            part.tofile(fh)

    # -- validity and fill
    def _get_valid(self):
        return self._valid

    def _set_valid(self, valid):
        self._valid = bool(valid)

    valid = property(lambda self: self._get_valid(),
                     lambda self, v: self._set_valid(v),
                     doc="Whether the frame holds usable data.")

    def _get_fill(self):
        return self._fill_value

    def _set_fill(self, value):
        self._fill_value = value

    fill_value = property(lambda self: self._get_fill(),
                          lambda self, v: self._set_fill(v),
                          doc="Value returned for invalid data (default 0).")

    def _fill(self, shape):
        kind = torch.complex64 if self.dtype.kind == 'c' else torch.float32
        return torch.full(tuple(shape), self.fill_value, dtype=kind, device='cuda')

    # -- array-like view of the decoded samples
    sample_shape = _delegate('payload', 'sample_shape')
    dtype = _delegate('payload', 'dtype')

    def __len__(self):
        return len(self.payload)

    @property
    def shape(self):
        return (len(self),) + tuple(self.sample_shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    ndim = property(lambda self: len(self.shape))
    nbytes = property(lambda self: self.header.nbytes + self.payload.nbytes)

    def __getitem__(self, item=()):
        if isinstance(item, str):               # header field
            return self.header[item]
        if self.valid:
            return self.payload[item]
        # invalid: only the SHAPE of the answer depends on the item
        return self._fill(np.empty(self.shape, dtype=bool)[item].shape)

    def __setitem__(self, item, value):
        """Header key -> header; samples -> payload (base/frame.py:203-207)."""
        if isinstance(item, str):
            self.header[item] = value
        else:
            self.payload[item] = value

    data = property(__getitem__, doc="Full decoded frame (device tensor).")

    def __array__(self, dtype=None, copy=None):
        """Decoded frame on the host (base/frame.py:182-187): the one place
        where a frame's samples leave the device."""
        host = to_numpy(self.data)
        return host if dtype in (None, host.dtype) else host.astype(dtype)

    # -- header passthrough
    def keys(self):
        return self.header.keys()

    def __contains__(self, key):
        return key in self.keys()

    def __getattr__(self, attr):
        # anything the frame does not define itself (time, bps, ...) is looked
        # up on the header; guard against recursion before __init__ ran
        if attr in ('header', 'payload') or attr.startswith('__'):
            raise AttributeError(attr)
        try:
            return getattr(self.header, attr)
        except AttributeError:
            raise AttributeError("{} object has no attribute {}"
                                 .format(type(self).__name__, attr)) from None

    def __setattr__(self, attr, value):
        # the header's settable properties are set on the header
        # (base/frame.py:228-234: ``frame.sample_rate = ...``, ``frame.time = ...``)
        if attr not in ('header', 'payload', 'valid'):
            header = self.__dict__.get('header')
            if header is not None and attr in getattr(type(header), '_properties', ()):
                return setattr(header, attr, value)
        return super().__setattr__(attr, value)

    def __eq__(self, other):
        return (type(other) is type(self)
                and (self.valid, self.header, self.payload)
                == (other.valid, other.header, other.payload))

    __hash__ = None


def block_frame_class(name, header_class, payload_class, doc):
    """Frame type for block formats whose frames are always valid and whose
    payloads are memory mapped (GUPPI, DADA): header + payload, nothing else."""

    def fromfile(cls, fh, memmap=True, verify=True):
        header = header_class.fromfile(fh, verify=verify)
        return cls(header, payload_class.fromfile(fh, header=header, memmap=memmap),
                   verify=verify)

    def fromdata(cls, data, header=None, *, valid=None, verify=True, **kwargs):
        if header is None:              # (header from the keywords: base/frame.py:114-134 in the reference)
            header = header_class.fromvalues(verify=verify, **kwargs)
        return cls(header, payload_class.fromdata(data, header=header), valid=valid, verify=verify)

    return type(name, (FrameBase,), dict(
        __doc__=doc, _header_class=header_class, _payload_class=payload_class,
        fromfile=classmethod(fromfile), fromdata=classmethod(fromdata)))
