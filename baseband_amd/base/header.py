"""Bit-field headers: a compact table-driven equivalent of the reference's
``HeaderParser``/``VLBIHeaderBase`` (base/header.py:35-87,250-667).

A header class lists its fields as ``name -> (word, bit, nbits[, default])``;
``header['name']`` extracts ``(words[word] >> bit) & (2**nbits - 1)`` (bool
for single bits, two words for nbits == 64), which is exactly what the
reference's generated parsers do.
"""
import struct

import numpy as np

four_word_struct = struct.Struct('<4I')
eight_word_struct = struct.Struct('<8I')


class HeaderParser(dict):
    """Field table of a header: ``name -> (word, bit, nbits[, default])`` in the
    order given -- the reference's class of the same name as far as header
    DEFINITIONS use it (base/header.py:122-286; docs/tutorials/new_edv.rst):
    made from a tuple of ``(name, (word, bit, nbits[, default]))`` pairs or a
    mapping, joined with ``|`` (or ``+``, the spelling before baseband 4.0) to
    another `HeaderParser`, with `defaults`, and `parsers` / `setters`: functions
    that take the field out of, or put it into, a sequence of 32-bit words.
    Entries of 64 bits span two words."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.update(*args, **kwargs)

    @staticmethod
    def _checked(name, spec):
        spec = tuple(spec)
        if not 3 <= len(spec) <= 4:
            raise ValueError("header entry {!r} needs (word, bit, nbits[, default])".format(name))
        word, bit, nbits = spec[:3]
        if not (nbits == 64 and bit == 0) and (nbits < 1 or bit + nbits > 32):
            raise ValueError("header entry {!r}: {} bits from bit {} do not fit a 32-bit word "
                             "(64 bits: two whole words)".format(name, nbits, bit))
        return spec

    def __setitem__(self, name, spec):
        super().__setitem__(name, self._checked(name, spec))

    def update(self, *args, **kwargs):
        try:
            items = dict(*args, **kwargs)
        except (TypeError, ValueError):
            raise ValueError("a header parser is updated from pairs of (name, (word, bit, nbits[, default])) "
                             "or a mapping") from None
        for name, spec in items.items():
            self[name] = spec

    def __or__(self, other):
        if not isinstance(other, HeaderParser):
            raise TypeError("can only join a HeaderParser with another HeaderParser, not {}"
                            .format(type(other).__name__))
        new = type(self)(self)
        new.update(other)
        return new

    __add__ = __or__

    def __ior__(self, other):
        if not isinstance(other, HeaderParser):
            raise TypeError("can only join a HeaderParser with another HeaderParser")
        self.update(other)
        return self

    def copy(self):
        return type(self)(self)

    @property
    def defaults(self):
        """name -> default value (None where an entry has none)."""
        return {name: (spec[3] if len(spec) > 3 else None) for name, spec in self.items()}

    @staticmethod
    def _parser(spec):
        word, bit, nbits = spec[:3]
        if nbits == 64:
            return lambda words: int(words[word]) + (int(words[word + 1]) << 32)
        mask = (1 << nbits) - 1
        if nbits == 1:
            return lambda words: bool((int(words[word]) >> bit) & 1)
        return lambda words: (int(words[word]) >> bit) & mask

    @staticmethod
    def _setter(spec):
        word, bit, nbits = spec[:3]
        default = spec[3] if len(spec) > 3 else None

        def setter(words, value):
            if value is None and default is not None:
                value = default
            value = int(value)
            if nbits == 64:
                words[word], words[word + 1] = value & 0xffffffff, value >> 32
            else:
                mask = (1 << nbits) - 1
                if value & mask != value:
                    raise ValueError("{0} cannot be represented with {1} bits".format(value, nbits))
                words[word] = (int(words[word]) & ~(mask << bit) & 0xffffffff) | (value << bit)
            return words
        return setter

    @property
    def parsers(self):
        """name -> function(words) giving the field's value."""
        return {name: self._parser(spec) for name, spec in self.items()}

    @property
    def setters(self):
        """name -> function(words, value) putting the value in (None: the default)."""
        return {name: self._setter(spec) for name, spec in self.items()}

    def __repr__(self):
        return "{}({})".format(type(self).__name__, tuple(self.items()))


class BitFieldHeader:
    """Header made of 32-bit little-endian words with named bit fields."""

    _fields = {}            # name -> (word, bit, nbits, default)
    _struct = None
    _invariants = set()
    _stream_invariants = set()

    def __init__(self, words=None, verify=True, **kwargs):
        if words is None:
            self.words = [0] * (self._struct.size // 4)
            self._mutable = True
        else:
            # as the reference's headers, which keep the caller's container: a header
            # of a list or a writeable array can be changed, one of a tuple (what
            # `fromfile` unpacks) or a read-only array cannot (base/header.py:337-363
            # there).  The words are COPIED here: changes do not write through.
            self._mutable = (isinstance(words, list)
                             or bool(getattr(getattr(words, 'flags', None), 'writeable', False)))
            self.words = (list if self._mutable else tuple)(int(w) for w in words)
        if verify:
            self.verify()

    # -- dict-like access
    def keys(self):
        return self._fields.keys()

    def __contains__(self, key):
        return key in self._fields

    def __getitem__(self, key):
        try:
            word, bit, nbits = self._fields[key][:3]
        except KeyError:
            raise KeyError("{0} header does not contain {1}"
                           .format(self.__class__.__name__, key))
        if nbits == 64:
            # (an unsigned 64-bit NumPy integer, as the reference's word arithmetic gives)
            return np.uint64(int(self.words[word]) + (int(self.words[word + 1]) << 32))
        v = (self.words[word] >> bit) & ((1 << nbits) - 1)
        return bool(v) if nbits == 1 else v

    def __setitem__(self, key, value):
        if not self._mutable:
            raise TypeError("header is immutable; use .copy() to get a "
                            "mutable one.")
        word, bit, nbits = self._fields[key][:3]
        default = self._fields[key][3] if len(self._fields[key]) > 3 else None
        mask = (1 << nbits) - 1
        if value is None:
            if default is None:
                raise ValueError("no default value so cannot set to 'None'.")
            value = default
        elif value is True:
            value = mask                 # all bits: used for invariant masks
        else:
            if isinstance(value, np.ndarray):       # (a one-element array, e.g. packed bits viewed as '>u8')
                value = value.reshape(-1)[0]
            value = int(value)
            if value & mask != value:
                raise ValueError("{0} cannot be represented with {1} bits"
                                 .format(value, nbits))
        if nbits == 64:
            self.words[word] = value & 0xffffffff
            self.words[word + 1] = value >> 32
        else:
            w = self.words[word]
            self.words[word] = (w & ~(mask << bit) & 0xffffffff) | (value << bit)

    # property-like keywords `update` knows, applied in this order after the
    # plain header keys (base/header.py:420-450)
    _properties = ()

    @classmethod
    def fromkeys(cls, *args, verify=True, **kwargs):
        """Header from explicit values for header KEYS only (no derived
        properties, no defaults for missing keys: base/header.py:396-418)."""
        self = cls(None, *args, verify=False)
        missing = [k for k in self.keys() if k not in kwargs]
        if missing or len(kwargs) != len(list(self.keys())):
            raise KeyError("need keyword arguments for all keys in header "
                           "(missing or extra: {})".format(
                               sorted(set(missing) | (set(kwargs) - set(self.keys())))))
        for key, value in kwargs.items():
            self[key] = value
        if verify:
            self.verify()
        return self

    def update(self, *, verify=True, **kwargs):
        """Set header keys first, then derived properties in `_properties`
        order; anything left over draws a warning (base/header.py:420-450)."""
        import warnings
        for key in [k for k in kwargs if k in self.keys()]:
            self[key] = kwargs.pop(key)
        for key in self._properties:
            if key in kwargs:
                setattr(self, key, kwargs.pop(key))
        if kwargs:
            warnings.warn("some keywords unused in header update: {0}".format(kwargs))
        if verify:
            self.verify()

    def copy(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__dict__)
        new.words = list(self.words)
        new._mutable = True
        return new

    @property
    def mutable(self):
        return self._mutable

    @mutable.setter
    def mutable(self, mutable):
        self.words = (list if mutable else tuple)(self.words)
        self._mutable = bool(mutable)

    @property
    def nbytes(self):
        return self._struct.size

    def tofile(self, fh):
        return fh.write(self._struct.pack(*self.words))

    def verify(self):
        pass

    def invariants(self):
        """Keys whose bits are the same for all headers of this stream."""
        return self._stream_invariants

    def invariant_pattern(self, invariants=None):
        """(pattern words, mask words): bits shared by all headers of the
        stream (base/header.py:588-638)."""
        if invariants is None:
            invariants = self.invariants()
        if not invariants:
            raise ValueError("cannot create an invariant_mask without "
                             "some invariants")
        mask = [0] * len(self.words)
        for key in invariants:
            word, bit, nbits = self._fields[key][:3]
            if nbits == 64:
                mask[word] = mask[word + 1] = 0xffffffff
            else:
                mask[word] |= ((1 << nbits) - 1) << bit
        return list(self.words), mask

    def __eq__(self, other):
        return (type(self) is type(other)
                and list(self.words) == list(other.words))

    def __repr__(self):
        """Keys and values; BCD, CRC and sync-pattern fields in hex (base/header.py:497-500,661-667)."""
        def show(key, value):
            if key.startswith(('bcd', 'crc', 'sync_pattern')):
                try:
                    return hex(int(value))
                except Exception:
                    pass
            return str(value)
        name = self.__class__.__name__
        return ("<{0} {1}>".format(name, (",\n  " + len(name) * " ").join(
            ["{0}: {1}".format(k, show(k, self[k])) for k in self.keys()])))


def strided_header_words(buf, frame_nbytes, nwords, offset=0):
    """(nframes, nwords) uint32 view of the headers of a fixed-stride file
    image held in a uint8 NumPy array or memmap (no copy).  Only frames whose
    header lies completely inside the buffer are included."""
    if hasattr(buf, 'header_words'):         # helpers.sequentialfile.SequenceImage
        return buf.header_words(frame_nbytes, nwords, offset)
    buf = np.asarray(buf)[offset:]
    nframes = (len(buf) - 4 * nwords) // frame_nbytes + 1 if len(buf) >= 4 * nwords else 0
    u4 = np.frombuffer(buf, dtype='<u4', count=len(buf) // 4)
    return np.lib.stride_tricks.as_strided(
        u4, shape=(nframes, nwords), strides=(frame_nbytes, 4), writeable=False)


# the reference's names for this role (base/header.py:488-800)
ParsedHeaderBase = VLBIHeaderBase = BitFieldHeader
