"""Payload container: encoded words on the host, decode on the GPU.

Keeps the public surface of the reference's ``PayloadBase``
(base/payload.py:18-360) -- constructor arguments, ``fromfile``/``fromdata``,
``nbytes, shape, size, ndim, dtype, sample_shape, data``, ``len()``, item
access and ``__array__`` -- but is organised around one primitive: *decode
the smallest run of whole words that covers samples [start, stop)*.  The run
is expanded by the libbbdecode flat kernel (``bb_decode_frames``) from a copy
of the words that is uploaded to HBM once; everything that comes back is a
device tensor (float32 or complex64).  ``__array__`` is the only place where
samples travel back to the host.
"""
import operator

import numpy as np
import torch

from .. import kernels
from ..staging import to_numpy


def _resolve_index(index, length, who):
    """Normalise an int or slice over `length` samples.

    Returns (start, stop, step, scalar).  Only forward steps are supported,
    like in the reference (base/payload.py:262-264)."""
    if isinstance(index, slice):
        start, stop, step = index.indices(length)
        assert step > 0, "cannot deal with negative steps yet."
        return start, max(stop, start), step, False
    try:
        i = operator.index(index)
    except Exception:
        raise TypeError("{0} object can only be indexed or sliced.".format(who))
    if i < 0:
        i += length
    if i < 0 or i >= length:
        raise IndexError("{0} index out of range.".format(who))
    return i, i + 1, 1, True


class RowSetMixin:
    """``payload[item] = data`` for payloads that decode by rows of complete
    samples (GUPPI, DADA): rows [lo, hi) that cover the item -- whole
    `_row_granule` blocks -- are decoded on the GPU, updated, packed again by
    the GPU encoder and stored with ``_store_rows`` (guppi/payload.py:112-140,
    base/payload.py:332-347)."""
    _row_granule = 1

    def __setitem__(self, item, data):
        if isinstance(item, tuple):
            sample_index = item[1:]
            first = item[0] if item else slice(None)
        else:
            sample_index, first = (), item
        start, stop, step, scalar = _resolve_index(first, len(self), type(self))
        if stop == start:
            return
        g = self._row_granule
        lo, hi = start // g * g, min(-(-stop // g) * g, len(self))
        if not isinstance(data, torch.Tensor):
            data = torch.from_numpy(np.ascontiguousarray(data))
        data = data.to('cuda')
        want = torch.complex64 if self.complex_data else torch.float32
        if (not scalar and step == 1 and not sample_index and (lo, hi) == (start, stop)
                and tuple(data.shape) == (stop - start,) + tuple(self.sample_shape)):
            block = data.to(want)               # whole rows: nothing to keep
        else:
            self._dwords = None                 # the words may have changed
            block = self[lo:hi].clone()
            if scalar:
                index = (start - lo,) + sample_index
            else:
                index = (slice(start - lo, stop - lo, step),) + sample_index
            _assign(block, index, data)
        self._store_rows(lo, hi, block)
        self._dwords = None


def _assign(block, index, values):
    """``block[index] = values`` with NumPy's error for shapes that do not broadcast
    (a ValueError; torch raises RuntimeError), as callers of the reference expect."""
    target = tuple(block[index].shape)
    given = tuple(values.shape)
    while len(given) > len(target) and given[0] == 1:       # (leading axes of length 1 do not count)
        given = given[1:]
    try:
        fits = tuple(torch.broadcast_shapes(given, target)) == target
    except RuntimeError:
        fits = False
    if not fits and target == () and block.is_complex():
        # (NumPy's wording and class for a sequence into one complex element)
        raise TypeError("only length-1 arrays can be converted to Python scalars")
    if not fits:            # (before any cast: NumPy complains about the shape first, and only that)
        raise ValueError("could not broadcast input array from shape {} into shape {}".format(
            tuple(values.shape), target))
    block[index] = values.reshape(given).to(block.dtype)


class PayloadBase:
    # subclass knobs -------------------------------------------------------
    _nbytes = None                      # fixed payload size, if the format has one
    _memmap = False                     # map rather than read in fromfile
    _dtype_word = np.dtype('<u4')       # dtype the encoded words must have
    _coder_id = None                    # enum bb_coder handed to the kernels
    _sample_shape_maker = None          # namedtuple factory for sample_shape

    def __init__(self, words, *, header=None, sample_shape=(), bps=2,
                 complex_data=False):
        if header is not None:
            sample_shape, bps = header.sample_shape, header.bps
            complex_data = header.complex_data
            expected = header.payload_nbytes
            if self._nbytes is not None and self._nbytes != expected:
                raise ValueError("header payload size should be {0}"
                                 .format(self._nbytes))
            self._nbytes = expected
        if words.dtype != self._dtype_word:
            raise ValueError("encoded data should have dtype {0}"
                             .format(self._dtype_word))
        if self._nbytes not in (None, words.nbytes):
            raise ValueError("encoded data should have length {0}"
                             .format(self._nbytes))
        maker = self._sample_shape_maker
        self.sample_shape = maker(*sample_shape) if maker else tuple(sample_shape)
        self.words, self.bps, self.complex_data = words, bps, complex_data
        self._sample_size = int(np.prod(sample_shape, dtype=np.int64)) if len(sample_shape) else 1
        # bits per complete sample; subclasses may adjust (VDIF 3-bit etc.)
        self._bpfs = bps * (2 if complex_data else 1) * self._sample_size
        self._coder = bps               # key of the reference's _decoders dict
        self._dwords = None             # device copy of the words (lazy)

    # construction -----------------------------------------------------------
    @staticmethod
    def _read_words(fh, nbytes, dtype, memmap):
        if not memmap:
            raw = fh.read(nbytes)
            if len(raw) < nbytes:
                raise EOFError("could not read full payload.")
            return np.frombuffer(raw, dtype=dtype)
        count = nbytes // dtype.itemsize
        if hasattr(fh, 'memmap'):
            return fh.memmap(dtype=dtype, shape=(count,))
        start = fh.tell()
        words = np.memmap(fh, mode=fh.mode.replace('b', ''), dtype=dtype,
                          offset=start, shape=(count,))
        fh.seek(start + words.nbytes)
        return words

    @classmethod
    def fromfile(cls, fh, header=None, *, payload_nbytes=None, dtype=None,
                 memmap=None, **kwargs):
        """Take ``payload_nbytes`` (from the header, the argument, or the
        class) bytes from `fh` as payload words (base/payload.py:84-139)."""
        if header is not None:
            kwargs['header'] = header
            payload_nbytes = header.payload_nbytes
        if payload_nbytes is None:
            payload_nbytes = cls._nbytes
        if payload_nbytes is None:
            raise ValueError(
                "payload_nbytes or header should be passed in "
                "if no default payload size is defined on the class.")
        words = cls._read_words(fh, payload_nbytes,
                                cls._dtype_word if dtype is None else dtype,
                                cls._memmap if memmap is None else memmap)
        self = cls(words, **kwargs)
        # a file reader of this package lends payloads read one after the other
        # their bytes from a window of the file kept in HBM (FileBase._lend_device_words)
        lend = getattr(fh, '_lend_device_words', None)
        if lend is not None:
            lend(self)
        return self

    @classmethod
    def fromdata(cls, data, header=None, bps=2, **kwargs):
        """Encode samples as a payload on the GPU (bb_encode_flat; bit-identical
        to the reference encoders, base/payload.py:141-188).  Host arrays are
        uploaded first: there is no CPU encoder on this path."""
        data = kernels.as_device_samples(data)
        is_complex = data.is_complex()
        if header is None:
            words = cls._encode_device(data, bps, **kwargs)
            return cls(words, bps=bps, sample_shape=tuple(data.shape[1:]),
                       complex_data=is_complex)
        if tuple(header.sample_shape) != tuple(data.shape[1:]):
            raise ValueError(
                f"header is for sample_shape={header.sample_shape} "
                f"but data has {data.shape[1:]}")
        if bool(header.complex_data) != is_complex:
            kinds = ['complex' if c else 'real'
                     for c in (header.complex_data, is_complex)]
            raise ValueError("header is for {0} data but data are {1}".format(*kinds))
        return cls(cls._encode_device(data, header.bps, **kwargs), header=header)

    @classmethod
    def _encode_device(cls, data, bps, coder_id=None, **kwargs):
        """Pack a device tensor with the GPU encoder of this payload's coder."""
        coder_id = cls._coder_id if coder_id is None else coder_id
        try:
            packed = kernels.encode_flat(data, coder_id, bps)
        except KeyError:
            raise ValueError(f"{cls.__name__} cannot encode data with {bps} bits") from None
        return packed.cpu().numpy().view(cls._dtype_word)

    def tofile(self, fh):
        return fh.write(self.words.tobytes())

    # geometry -----------------------------------------------------------------
    nbytes = property(lambda self: self.words.nbytes,
                      doc="Size of the payload in bytes.")
    shape = property(lambda self: (len(self),) + tuple(self.sample_shape),
                     doc="Shape of the decoded data.")
    size = property(lambda self: len(self) * self._sample_size,
                    doc="Number of component samples in the decoded data.")
    ndim = property(lambda self: 1 + len(self.sample_shape))
    dtype = property(lambda self: np.dtype('c8' if self.complex_data else 'f4'),
                     doc="NumPy dtype the decoded data corresponds to.")

    def __len__(self):
        return 8 * self.words.nbytes // self._bpfs

    # item access ----------------------------------------------------------------
    def _covering_words(self, start, stop):
        """Smallest [w0, w1) word range holding samples [start, stop) and the
        number of decoded samples that precede `start` in it.  Samples either
        tile words or words tile samples (base/payload.py:283-310)."""
        bpw, bpfs = 8 * self.words.itemsize, self._bpfs
        if bpfs % bpw and bpw % bpfs:
            raise TypeError("do not know how to extract data when full "
                            "samples have {0} bits and words have {1} bits"
                            .format(bpfs, bpw))
        w0 = start * bpfs // bpw
        w1 = -(-stop * bpfs // bpw)
        return w0, w1, start - w0 * bpw // bpfs

    def _item_to_slices(self, item):
        """(word slice, data index) pair: decoding ``words[word slice]`` and
        applying ``data index`` gives ``item``.  Same contract as the
        reference's helper of this name (base/payload.py:226-312)."""
        rest = ()
        if isinstance(item, tuple):
            item, rest = (item[0], item[1:]) if item else (slice(None), ())
        start, stop, step, scalar = _resolve_index(item, len(self),
                                                   type(self))
        if stop - start == len(self) and not scalar:
            return slice(None), (slice(None, None, None if step == 1 else step),) + rest
        w0, w1, lead = self._covering_words(start, stop)
        if scalar:
            return slice(w0, w1), (lead,) + rest
        tail_open = (stop * self._bpfs) % (8 * self.words.itemsize) == 0
        index = slice(lead or None, None if tail_open else lead + stop - start,
                      None if step == 1 else step)
        return slice(w0, w1), (index,) + rest

    def _device_words(self):
        """Payload bytes in HBM.  The upload is kept only for read-only words
        (bytes read from a file, read-only mappings): writable words can be
        changed behind our back (a writable ``memmap_frame``, an array shared
        with another payload), and the reference decodes the CURRENT words on
        every access (base/payload.py:327-330)."""
        if self._dwords is not None:
            return self._dwords
        dwords = kernels.to_device_bytes(self.words)
        if not getattr(self.words, 'flags', None) or not self.words.flags.writeable:
            self._dwords = dwords
        return dwords

    def _decode(self, byte_start, byte_stop):
        """Flat float32 device tensor for payload bytes [byte_start,
        byte_stop): the seam where the reference calls
        ``self._decoders[self._coder](words)`` (base/payload.py:314-315)."""
        if self._coder_id is None:
            raise KeyError(self._coder)
        if byte_stop == byte_start:
            return torch.empty(0, dtype=torch.float32, device='cuda')
        return kernels.decode_frames(self._device_words(), 1,
                                     byte_stop - byte_start, self._coder_id,
                                     self.bps, src0=byte_start)

    def _as_dtype(self, flat):
        return torch.view_as_complex(flat.view(-1, 2)) if self.complex_data else flat

    def __getitem__(self, item=()):
        word_slice, index = self._item_to_slices(item)
        w0, w1, _ = word_slice.indices(len(self.words))
        size = self.words.itemsize
        block = self._as_dtype(self._decode(w0 * size, w1 * size))
        return block.reshape(-1, *self.sample_shape)[index]

    def _encode(self, data):
        """Device samples ``(n,) + sample_shape`` -> packed words (host
        array): the seam where the reference calls
        ``self._encoders[self._coder](data)`` (base/payload.py:317-325)."""
        if self._coder_id is None:
            raise ValueError("{} cannot encode data with {} bits"
                             .format(type(self).__name__, self.bps))
        try:
            packed = kernels.encode_flat(data, self._coder_id, self.bps)
        except KeyError:
            raise ValueError("{} cannot encode data with {} bits"
                             .format(type(self).__name__, self.bps)) from None
        return packed.cpu().numpy().view(self._dtype_word)

    def _fresh_block(self, w0, w1):
        """Decoded copy of words [w0, w1) as (n,) + sample_shape, taken from
        the host words as they are now."""
        self._dwords = None
        size = self.words.itemsize
        block = self._as_dtype(self._decode(w0 * size, w1 * size))
        return block.reshape(-1, *self.sample_shape).clone()

    def __setitem__(self, item, data):
        """Replace samples: the covering words are decoded on the GPU, the
        new values inserted, and the block packed again by the GPU encoder
        (base/payload.py:332-347).  The words must be writable (made by
        ``fromdata`` or memory-mapped for writing)."""
        word_slice, index = self._item_to_slices(item)
        w0, w1, _ = word_slice.indices(len(self.words))
        if not isinstance(data, torch.Tensor):
            data = torch.from_numpy(np.ascontiguousarray(data))
        data = data.to('cuda')
        ns = len(self.sample_shape)
        nrows = (w1 - w0) * self.words.itemsize * 8 // self._bpfs      # complete samples in those words
        whole = (index == (slice(None),) and data.ndim == ns + 1
                 and tuple(data.shape) == (nrows,) + tuple(self.sample_shape)
                 and data.is_complex() == self.complex_data)
        if not whole:
            block = self._fresh_block(w0, w1)
            _assign(block, index, data)
            data = block
        encoded = self._encode(kernels.as_device_samples(data))
        self.words[w0:w1] = encoded.reshape(-1)
        self._dwords = None

    data = property(__getitem__, doc="Full decoded payload (device tensor).")

    def __array__(self, dtype=None, copy=None):
        host = to_numpy(self.data)
        return host if dtype in (None, host.dtype) else host.astype(dtype)

    def __eq__(self, other):
        if type(other) is not type(self):
            return False
        same_meta = (self.shape, self.dtype) == (other.shape, other.dtype)
        if not same_meta:
            return False
        if self.words is other.words:
            return True
        # (byte for byte: one may hold its words as int8 samples, the other as dwords)
        a = np.ascontiguousarray(self.words).reshape(-1).view(np.uint8)
        b = np.ascontiguousarray(other.words).reshape(-1).view(np.uint8)
        return a.size == b.size and bool(np.array_equal(a, b))

    __hash__ = None
