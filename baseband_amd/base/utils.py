"""Small host-side helpers with the names of the reference's
``baseband.base.utils`` (base/utils.py:13-250): least common multiple, binary
coded decimals, byte patterns and the two kinds of cyclic redundancy checks
(per value: Mark 5B headers; per bit stream: Mark 4 headers).  Headers are
host metadata -- the bulk checks of every frame in a file run in the scan
kernels (``bb_mark5b_locate`` checks the CRC-16 of each candidate header on
the GPU); these serve header construction and single-header verification."""
import operator
from math import gcd

import numpy as np

__all__ = ['lcm', 'bcd_decode', 'bcd_encode', 'byte_array', 'CRC', 'CRCStack']


def lcm(a, b):
    """Least common multiple of a and b."""
    return abs(a * b) // gcd(a, b)


def _digits(value, base):
    n = value.dtype.itemsize * 2
    place = np.arange(n)
    if base == 16:
        return (value[..., np.newaxis] >> (4 * place).astype(value.dtype)) & 0xf
    return (value[..., np.newaxis] // (10 ** place).astype(value.dtype)) % 10


def bcd_decode(value):
    """Binary coded decimal -> integer, e.g. 0x1234 -> 1234; arrays of integers
    element by element.  A nibble above 9 is a ValueError (base/utils.py:18-34)."""
    try:
        return int('{:x}'.format(operator.index(value)))
    except TypeError as exc:
        if getattr(getattr(value, 'dtype', None), 'kind', '') not in 'iu' or \
                getattr(value, 'dtype', None) is None:
            raise exc
    digits = _digits(value, 16)
    if digits.size and digits.max() > 9:
        bad = value[(digits > 9).any(-1)].ravel()[0]
        raise ValueError("invalid BCD encoded value {0}={1}.".format(bad, hex(bad)))
    place = np.arange(digits.shape[-1])
    return (digits.astype(np.int64) * 10 ** place).sum(-1)


def bcd_encode(value):
    """Integer -> binary coded decimal, e.g. 1234 -> 0x1234 (base/utils.py:37-49)."""
    try:
        return int('{:d}'.format(operator.index(value)), base=16)
    except TypeError as exc:
        if getattr(getattr(value, 'dtype', None), 'kind', '') not in 'iu' or \
                getattr(value, 'dtype', None) is None:
            raise exc
    digits = _digits(value, 10).astype(np.int64)
    place = np.arange(digits.shape[-1])
    return (digits << (4 * place)).sum(-1)


def byte_array(pattern):
    """Pattern -> array of bytes: arrays and ``bytes`` are viewed, (iterables
    of) unsigned 32-bit integers are stored little-endian (base/utils.py:52-76)."""
    if isinstance(pattern, (np.ndarray, bytes)):
        return np.atleast_1d(pattern).view('u1')
    pattern = np.array(pattern, ndmin=1)
    if (pattern.dtype.kind not in 'uif' or pattern.min() < 0
            or pattern.max() >= 1 << 32):
        raise ValueError('values have to fit in 32 bit unsigned int.')
    return pattern.astype('<u4').view('u1')


class CRC:
    """Cyclic redundancy check with the given binary-encoded divisor, e.g.
    0x18005 (x^16 + x^15 + x^2 + 1) for Mark 5B headers.  Calling the instance
    gives the CRC of an integer (of any length) or of every element of an
    array; ``check`` tells whether a value that ends in its CRC is consistent
    (base/utils.py:93-197)."""

    def __init__(self, polynomial):
        self.polynomial = operator.index(polynomial)

    def __len__(self):
        return self.polynomial.bit_length() - 1

    def __call__(self, stream):
        return self._remainder(stream, extend=True)

    def check(self, stream):
        return self._remainder(stream, extend=False) == 0

    def _remainder(self, stream, extend):
        try:
            value = operator.index(stream)
        except TypeError:
            return self._remainder_array(stream, extend)
        if extend:
            value <<= len(self)
        width = self.polynomial.bit_length()
        while value.bit_length() >= width:
            value ^= self.polynomial << (value.bit_length() - width)
        return value

    def _remainder_array(self, array, extend):
        array = np.array(array, copy=True, dtype='u8')
        if extend:
            array <<= np.uint64(len(self))
        width = self.polynomial.bit_length()
        top = int(array.max()).bit_length() if array.size else 0
        while top >= width:
            # divide at bit `top` wherever it is set
            hit = (array >> np.uint64(top - 1)) & np.uint64(1)
            array ^= hit * np.uint64(self.polynomial << (top - width))
            top -= 1
        return array


class CRCStack(CRC):
    """CRC over bit streams stacked in the bits of unsigned integers (or a
    single stream of bool): element i of the array holds bit i of every
    stream, as the tracks of a Mark 4 header do (base/utils.py:200-248)."""

    def check(self, stream):
        return np.all(self._remainder(stream, extend=False) == 0)

    def _remainder(self, stream, extend):
        ncrc = len(self)
        stream = np.asarray(stream)
        if extend:
            stream = np.hstack((stream, np.zeros(ncrc, stream.dtype)))
        else:
            stream = stream.copy()
        taps = np.array([-int(bit) for bit in '{:b}'.format(self.polynomial)],
                        dtype='i1').astype(stream.dtype)
        for i in range(len(stream) - ncrc):
            stream[i:i + ncrc + 1] ^= stream[i] & taps
        return stream[-ncrc:]


def named_sample_shape(unsliced, fields, squeeze=False, subset=()):
    """The shape of a complete sample as the reference's streams give it
    (base/base.py:460-485,719-775 there): a named tuple ``SampleShape`` over `fields`
    (``nthread, nchan`` for VDIF, ``npol, nchan`` for DADA / GUPPI, ``nchan`` ...), with
    the fields of length 1 dropped when squeezed, and -- after a `subset` -- the names of
    the dimensions that are left when indexing each dimension by itself says so; a plain
    tuple otherwise (advanced indexing across dimensions, nothing left, no names)."""
    from collections import namedtuple
    import numpy as np
    shape = tuple(int(d) for d in unsliced)
    if not fields or len(fields) != len(shape):
        fields = None
    if squeeze:
        if fields is not None:
            fields = [f for f, d in zip(fields, shape) if d > 1]
        shape = tuple(d for d in shape if d > 1)
    if subset:
        subset = tuple(subset)
        try:
            # the subset has to pick from each sample alone: in range, something left, and
            # no mixing of the sample axis into the rest (advanced indices that broadcast
            # against it), as the reference checks on a dummy sample (base/base.py:727-743)
            probe = np.empty((2,) + shape, dtype=bool)[(slice(None),) + subset]
            assert 0 not in probe.shape and probe.shape[:1] == (2,)
        except (IndexError, AssertionError) as exc:
            exc.args += ("subset {} cannot be used to properly index {}samples with shape {}."
                         .format(subset, "squeezed " if squeeze else "", shape),)
            raise exc
        full = probe.shape[1:]
        if fields is None or full == () or len(subset) > len(shape):
            return tuple(full)
        kept, axis = [], 0
        try:
            for field, dim, item in zip(fields, shape, subset + (slice(None),) * (len(shape) - len(subset))):
                left = np.empty(dim)[item].shape
                assert len(left) <= 1                   # (no multi-dimensional indexing of one axis)
                if len(left) == 1:
                    assert left[0] == full[axis]
                    kept.append(field)
                    axis += 1
            assert axis == len(full)
        except Exception:
            return tuple(full)
        shape, fields = tuple(full), kept
    if fields is None:
        return tuple(shape)
    if not fields:
        return namedtuple('SampleShape', [])()
    return namedtuple('SampleShape', ','.join(fields))(*shape)


class fixedvalue:
    """A value that is the same for every instance of a class, readable on the class
    too; setting it passes when the value is that value and raises ValueError otherwise
    (the reference's descriptor of the same name, base/utils.py:79-90: Mark 5B's
    ``payload_nbytes = 10000``, ``complex_data = False`` ...)."""

    def __init__(self, value, name=None):
        self.value, self.name = value, name

    def __set_name__(self, owner, name):
        self.name = name

    def __get__(self, instance, owner=None):
        return self.value

    def __set__(self, instance, value):
        if value != self.value:
            raise ValueError("'{}' can only be set to {}.".format(self.name, self.value))
