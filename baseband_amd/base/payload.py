"""Payload container: encoded words on the host, decode on the GPU.

Host-side mirror of the reference's ``PayloadBase`` (base/payload.py:18-360):
same constructor, ``fromfile``/``fromdata`` class methods, ``nbytes, shape,
size, ndim, dtype, sample_shape, data`` properties, ``__len__``,
``__getitem__`` and ``__array__``.  The difference is where ``_decode`` runs:
the words are uploaded once and expanded by the libbbdecode flat kernel
(bb_decode_frames), and ``data`` / ``__getitem__`` return device tensors
(torch, float32 or complex64).  ``__array__`` performs the device-to-host
copy for NumPy consumers.
"""
import operator
from functools import reduce

import numpy as np
import torch

from .. import kernels


class PayloadBase:
    # Possible fixed payload size in bytes.
    _nbytes = None
    _memmap = False
    _dtype_word = np.dtype('<u4')
    # ABI coder id (include/bbdecode.h enum bb_coder); set by subclasses.
    _coder_id = None
    # bits per sample the coder supports; anything else -> KeyError on decode,
    # like a missing key in the reference's _decoders dict.
    _sample_shape_maker = None

    def __init__(self, words, *, header=None, sample_shape=(), bps=2,
                 complex_data=False):
        if header is not None:
            sample_shape = header.sample_shape
            bps = header.bps
            complex_data = header.complex_data
            if self._nbytes is None:
                self._nbytes = header.payload_nbytes
            elif self._nbytes != header.payload_nbytes:
                raise ValueError("header payload size should be {0}"
                                 .format(self._nbytes))
        self.words = words
        if self._sample_shape_maker is not None:
            self.sample_shape = self._sample_shape_maker(*sample_shape)
        else:
            self.sample_shape = tuple(sample_shape)
        self._sample_size = reduce(operator.mul, sample_shape, 1)
        self.bps = bps
        self.complex_data = complex_data
        self._bpfs = bps * (2 if complex_data else 1) * self._sample_size
        self._coder = bps
        self._dwords = None
        if self._nbytes is not None and self._nbytes != words.nbytes:
            raise ValueError("encoded data should have length {0}"
                             .format(self._nbytes))
        if words.dtype != self._dtype_word:
            raise ValueError("encoded data should have dtype {0}"
                             .format(self._dtype_word))

    @classmethod
    def fromfile(cls, fh, header=None, *, payload_nbytes=None, dtype=None,
                 memmap=None, **kwargs):
        """Read payload words from a filehandle (base/payload.py:84-139)."""
        if header is not None:
            payload_nbytes = header.payload_nbytes
            kwargs['header'] = header
        elif payload_nbytes is None:
            payload_nbytes = cls._nbytes
            if payload_nbytes is None:
                raise ValueError(
                    "payload_nbytes or header should be passed in "
                    "if no default payload size is defined on the class.")
        if dtype is None:
            dtype = cls._dtype_word
        if memmap is None:
            memmap = cls._memmap
        if memmap:
            shape = (payload_nbytes // dtype.itemsize,)
            if hasattr(fh, 'memmap'):
                words = fh.memmap(dtype=dtype, shape=shape)
            else:
                mode = fh.mode.replace('b', '')
                offset = fh.tell()
                words = np.memmap(fh, mode=mode, dtype=dtype, offset=offset,
                                  shape=shape)
                fh.seek(offset + words.nbytes)
        else:
            s = fh.read(payload_nbytes)
            if len(s) < payload_nbytes:
                raise EOFError("could not read full payload.")
            words = np.frombuffer(s, dtype=dtype)
        return cls(words, **kwargs)

    def tofile(self, fh):
        return fh.write(self.words.tobytes())

    @classmethod
    def fromdata(cls, data, header=None, bps=2, **kwargs):
        """Encode data as a payload (host side; used to synthesise inputs)."""
        if isinstance(data, torch.Tensor):
            data = data.cpu().numpy()
        sample_shape = data.shape[1:]
        complex_data = data.dtype.kind == 'c'
        if header:
            bps = header.bps
            if tuple(header.sample_shape) != tuple(sample_shape):
                raise ValueError(
                    f"header is for sample_shape={header.sample_shape} "
                    f"but data has {sample_shape}")
            if header.complex_data != complex_data:
                raise ValueError("header is for {0} data but data are {1}"
                                 .format(*(('complex' if c else 'real') for c
                                           in (header.complex_data,
                                               complex_data))))
            base_kwargs = {"header": header}
        else:
            base_kwargs = {"bps": bps, "sample_shape": sample_shape,
                           "complex_data": complex_data}
        words = cls._encode_data(data, bps, **kwargs)
        return cls(words, **base_kwargs)

    @classmethod
    def _encode_data(cls, data, bps, **kwargs):
        raise ValueError(f"{cls.__name__} cannot encode data")

    # ----- array-like properties (base/payload.py:190-224)
    def __array__(self, dtype=None, copy=None):
        a = self.data.cpu().numpy()
        return a if dtype is None or dtype == a.dtype else a.astype(dtype)

    @property
    def nbytes(self):
        return self.words.nbytes

    def __len__(self):
        return self.words.nbytes * 8 // self._bpfs

    @property
    def shape(self):
        return (len(self),) + tuple(self.sample_shape)

    @property
    def size(self):
        return len(self) * self._sample_size

    @property
    def ndim(self):
        return 1 + len(self.sample_shape)

    @property
    def dtype(self):
        return np.dtype(np.complex64 if self.complex_data else np.float32)

    # ----- item -> minimal word range + residual slice (base/payload.py:226-312)
    def _item_to_slices(self, item):
        if isinstance(item, tuple):
            sample_index = item[1:]
            item = item[0] if item else slice(None)
        else:
            sample_index = ()
        nsample = len(self)
        is_slice = isinstance(item, slice)
        if is_slice:
            start, stop, step = item.indices(nsample)
            assert step > 0, "cannot deal with negative steps yet."
            n = stop - start
            if step == 1:
                step = None
        else:
            try:
                item = operator.index(item)
            except Exception:
                raise TypeError("{0} object can only be indexed or sliced."
                                .format(type(self)))
            if item < 0:
                item += nsample
            if not (0 <= item < nsample):
                raise IndexError("{0} index out of range.".format(type(self)))
            start, stop, step, n = item, item + 1, 1, 1

        if n == nsample:
            words_slice = slice(None)
            data_slice = slice(None, None, step) if is_slice else 0
        else:
            bpw = 8 * self.words.itemsize
            bpfs = self._bpfs
            if bpfs % bpw == 0:
                wpfs = bpfs // bpw
                words_slice = slice(start * wpfs, stop * wpfs)
                data_slice = slice(None, None, step) if is_slice else 0
            elif bpw % bpfs == 0:
                fspw = bpw // bpfs
                w_start, o_start = divmod(start, fspw)
                w_stop, o_stop = divmod(stop, fspw)
                words_slice = slice(w_start, w_stop + 1 if o_stop else w_stop)
                data_slice = slice(o_start if o_start else None,
                                   o_start + n if o_stop else None,
                                   step) if is_slice else o_start
            else:
                raise TypeError("do not know how to extract data when full "
                                "samples have {0} bits and words have {1} bits"
                                .format(bpfs, bpw))
        return words_slice, (data_slice,) + sample_index

    # ----- GPU decode
    def _device_words(self):
        """Payload bytes in HBM (uploaded once per payload)."""
        if self._dwords is None:
            self._dwords = kernels.to_device_bytes(self.words)
        return self._dwords

    def _decode(self, byte_start, byte_stop):
        """Decode words[byte_start:byte_stop] -> flat float32 device tensor.
        The seam the reference calls ``self._decoders[self._coder](words)``
        (base/payload.py:314-315)."""
        if self._coder_id is None:
            raise KeyError(self._coder)
        nbytes = byte_stop - byte_start
        if nbytes == 0:
            return torch.empty(0, dtype=torch.float32, device='cuda')
        return kernels.decode_frames(
            self._device_words(), 1, nbytes, self._coder_id, self.bps,
            src0=byte_start, src_stride=0)

    def _as_dtype(self, flat):
        if self.complex_data:
            flat = torch.view_as_complex(flat.view(-1, 2))
        return flat

    def __getitem__(self, item=()):
        words_slice, data_slice = self._item_to_slices(item)
        isz = self.words.itemsize
        w0, w1, _ = words_slice.indices(len(self.words))
        flat = self._decode(w0 * isz, w1 * isz)
        return self._as_dtype(flat).reshape(-1, *self.sample_shape)[data_slice]

    data = property(__getitem__, doc="Full decoded payload (device tensor).")

    def __eq__(self, other):
        return (type(self) is type(other)
                and self.shape == other.shape
                and self.dtype == other.dtype
                and (self.words is other.words
                     or np.all(self.words == other.words)))

    def __ne__(self, other):
        return not self.__eq__(other)
