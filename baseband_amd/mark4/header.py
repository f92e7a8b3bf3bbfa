"""Mark 4 headers (host side).

A Mark 4 header is 160 bits per tape track, stored bit-interleaved across the
first 160 stream words of a frame.  This mirrors ``Mark4Header``
(mark4/header.py:268-700): ``words`` is a ``(5, ntrack)`` uint32 array (one
column per track), fields come back as per-track arrays, and the derived
quantities (``fanout, bps, nchan, samples_per_frame, frame_nbytes, ...``)
follow mark4/header.py:540-650.  Times are ``numpy.datetime64[ns]``.
"""
import struct

import numpy as np
from ..base.utils import fixedvalue
from ..base.header import BitFieldHeader
from ..base.quantities import as_time

__all__ = ['Mark4Header', 'Mark4TrackHeader', 'stream2words', 'words2stream', 'MARK4_DTYPES',
           'PAYLOAD_NBITS', 'frame_header_streams']

MARK4_DTYPES = {8: '<u1', 16: '<u2', 32: '<u4', 64: '<u8'}
PAYLOAD_NBITS = 20000

_FIELDS = {
    'bcd_headstack1': (0, 0, 16, 0x3344),
    'bcd_headstack2': (0, 16, 16, 0x1122),
    'headstack_id': (1, 30, 2),
    'bcd_track_id': (1, 24, 6),
    'fan_out': (1, 22, 2),
    'magnitude_bit': (1, 21, 1),
    'lsb_output': (1, 20, 1),
    'converter_id': (1, 16, 4),
    'time_sync_error': (1, 15, 1, False),
    'internal_clock_error': (1, 14, 1, False),
    'processor_time_out_error': (1, 13, 1, False),
    'communication_error': (1, 12, 1, False),
    '_1_11_1': (1, 11, 1, False),
    '_1_10_1': (1, 10, 1, False),
    'track_roll_enabled': (1, 9, 1, False),
    'sequence_suspended': (1, 8, 1, False),
    'system_id': (1, 0, 8),
    '_1_0_1_sync': (1, 0, 1, 0),
    'sync_pattern': (2, 0, 32, 0xffffffff),
    'bcd_unit_year': (3, 28, 4),
    'bcd_day': (3, 16, 12),
    'bcd_hour': (3, 8, 8),
    'bcd_minute': (3, 0, 8),
    'bcd_second': (4, 24, 8),
    'bcd_fraction': (4, 12, 12),
    'crc': (4, 0, 12),
}

# Track assignments (tables 10-14 of Mark 4 memo 230.3), minus 2 so tracks
# start at 0; shape (fanout, nchan, bps) for 32 tracks
# (mark4/header.py:306-328).
_TRACK_ASSIGNMENTS = {
    (2, 4): np.array([[2, 10, 3, 11, 18, 26, 19, 27],
                      [4, 12, 5, 13, 20, 28, 21, 29],
                      [6, 14, 7, 15, 22, 30, 23, 31],
                      [8, 16, 9, 17, 24, 32, 25, 33]]).reshape(4, 4, 2) - 2,
    (1, 4): np.array([[2, 3, 10, 11, 18, 19, 26, 27],
                      [4, 5, 12, 13, 20, 21, 28, 29],
                      [6, 7, 14, 15, 22, 23, 30, 31],
                      [8, 9, 16, 17, 24, 25, 32, 33]]).reshape(4, 8, 1) - 2,
    (2, 2): np.array([[2, 6, 3, 7, 10, 14, 11, 15, 18, 22, 19, 23, 26, 30, 27, 31],
                      [4, 8, 5, 9, 12, 16, 13, 17, 20, 24, 21, 25, 28, 32, 29, 33]]
                     ).reshape(2, 8, 2) - 2,
    (1, 2): np.array([[2, 3, 6, 7, 10, 11, 14, 15, 18, 19, 22, 23, 26, 27, 30, 31],
                      [4, 5, 8, 9, 12, 13, 16, 17, 20, 21, 24, 25, 28, 29, 32, 33]]
                     ).reshape(2, 16, 1) - 2,
    (2, 1): np.array([[2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32,
                       3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 33]]
                     ).reshape(1, 16, 2) - 2,
}


def stream2words(stream, track=None):
    """Stream words (one bit per track) -> uint32 header words per track:
    bit `track` of stream word 32 j + i lands in bit 31 - i of word j
    (mark4/header.py:47-64)."""
    stream = np.asarray(stream)
    if track is None:
        track = np.arange(stream.dtype.itemsize * 8, dtype=stream.dtype)
    bits = ((stream.reshape(-1, 32, 1) >> track) & 1).astype(np.uint32)
    bits <<= np.arange(31, -1, -1, dtype=np.uint32).reshape(-1, 1)
    return np.bitwise_or.reduce(bits, axis=1)


def words2stream(words):
    """Inverse of `stream2words` (mark4/header.py:67-87)."""
    words = np.asarray(words, dtype=np.uint32)
    ntrack = words.shape[1]
    dtype = np.dtype(MARK4_DTYPES[ntrack])
    bit = np.arange(31, -1, -1, dtype=np.uint32).reshape(-1, 1)
    sel = ((words[:, np.newaxis, :] >> bit) & 1).astype(dtype)
    sel <<= np.arange(ntrack, dtype=dtype)
    return np.bitwise_or.reduce(sel, axis=2).astype(dtype).ravel()


def _bcd_decode_array(value):
    value = np.asarray(value).astype(np.int64)
    result = np.zeros_like(value)
    factor = 1
    v = value.copy()
    while np.any(v > 0):
        digit = v & 0xf
        if np.any(digit > 9):
            raise ValueError("invalid BCD encoded value")
        result += digit * factor
        factor *= 10
        v >>= 4
    return result


def _bcd_decode_int(value):
    """`_bcd_decode_array` for one number, in plain Python (a header's time at
    open(): 15 array decodes cost 1 ms of a 1.6 ms open)."""
    value, result, factor = int(value), 0, 1
    while value > 0:
        digit = value & 0xf
        if digit > 9:
            raise ValueError("invalid BCD encoded value")
        result += digit * factor
        factor *= 10
        value >>= 4
    return result


def _bcd_encode(value):
    result, shift, value = 0, 0, int(value)
    while value > 0:
        value, digit = divmod(value, 10)
        result += digit << shift
        shift += 4
    return result


def crc12_stream(stream):
    """CRC-12 (x^12 + x^11 + x^3 + x^2 + x + 1 = 0x180f, mark4/header.py:34-43)
    of a bit stream given as stream words (all tracks at once,
    base/utils.py:200-248)."""
    pol = [int(b) for b in '{:b}'.format(0x180f)]
    stream = np.concatenate([stream, np.zeros(12, stream.dtype)])
    ones = np.iinfo(stream.dtype).max
    pol_arr = np.array([ones if b else 0 for b in pol], dtype=stream.dtype)
    for i in range(len(stream) - 12):
        stream[i:i + 13] ^= stream[i] & pol_arr
    return stream[-12:]


class Mark4TrackHeader(BitFieldHeader):
    """Header of ONE Mark 4 track: five 32-bit words, the fields of memo 230.3
    (mark4/header.py:91-262 in the reference).  `Mark4Header` holds one such
    column per track; ``Mark4Header.track_header(k)`` gives track k's."""
    _fields = _FIELDS
    _struct = struct.Struct('<5I')
    _invariants = {'sync_pattern', '_1_0_1_sync'}
    _stream_invariants = _invariants | {'bcd_headstack1', 'bcd_headstack2', 'track_roll_enabled',
                                        'sequence_suspended', 'system_id'}
    _properties = ('decade', 'track_id', 'fraction', 'time')
    decade = None

    def __init__(self, words, decade=None, ref_time=None, verify=True):
        if decade is not None:
            self.decade = decade
        super().__init__(words, verify=verify)
        if decade is None and ref_time is not None:
            self.infer_decade(ref_time)

    def verify(self):
        assert len(self.words) == 5
        assert self['sync_pattern'] == 0xffffffff
        assert (self['bcd_fraction'] & 0xf) % 5 != 4
        if self.decade is not None:
            assert 1950 < self.decade < 3000
            assert self.decade % 10 == 0, "decade must end in zero"

    def infer_decade(self, ref_time):
        t = as_time(ref_time)
        year = int(t.astype('datetime64[Y]').astype(int)) + 1970
        frac = (t - np.datetime64(str(year), 'ns')) / np.timedelta64(365, 'D')
        self.decade = int(np.around(year + frac - int(self['bcd_unit_year']), decimals=-1))

    @property
    def track_id(self):
        return _bcd_decode_int(self['bcd_track_id'])

    @track_id.setter
    def track_id(self, track_id):
        self['bcd_track_id'] = _bcd_encode(int(track_id))

    @property
    def fraction(self):
        ms = _bcd_decode_int(self['bcd_fraction'])
        return (ms + (ms % 5) * 0.25) / 1000.

    @fraction.setter
    def fraction(self, fraction):
        ms = float(fraction) * 1000.
        if abs(ms / 1.25 - round(ms / 1.25)) > 1e-6:
            raise ValueError("{0} ms is not a multiple of 1.25 ms".format(ms))
        self['bcd_fraction'] = _bcd_encode(int(np.floor(ms + 1e-6)))

    def get_time(self):
        day = _bcd_decode_int(self['bcd_day'])
        sec = ((day - 1) * 24 + _bcd_decode_int(self['bcd_hour'])) * 3600 \
            + _bcd_decode_int(self['bcd_minute']) * 60 + _bcd_decode_int(self['bcd_second'])
        ns = sec * 10 ** 9 + int(round(self.fraction * 1e9))
        return (np.datetime64('{:04d}-01-01'.format(self.decade + int(self['bcd_unit_year'])), 'ns')
                + np.timedelta64(ns, 'ns'))

    def set_time(self, time):
        time = as_time(time)
        year = int(time.astype('datetime64[Y]').astype(int)) + 1970
        ns = int((time - np.datetime64('{:04d}-01-01'.format(year), 'ns')) / np.timedelta64(1, 'ns'))
        sec, rem = divmod(ns, 10 ** 9)
        self.fraction = rem / 1e9           # (first: it checks the 1.25 ms grid)
        day, sod = divmod(sec, 86400)
        hour, rest = divmod(sod, 3600)
        minute, second = divmod(rest, 60)
        self.decade = year // 10 * 10
        self['bcd_unit_year'] = year % 10
        self['bcd_day'] = _bcd_encode(day + 1)
        self['bcd_hour'] = _bcd_encode(hour)
        self['bcd_minute'] = _bcd_encode(minute)
        self['bcd_second'] = _bcd_encode(second)

    time = property(get_time, set_time)


class Mark4Header:
    """Decoder of a Mark 4 header containing all tracks."""
    _track_header = Mark4TrackHeader

    def track_header(self, track):
        """The header of one track as a `Mark4TrackHeader`."""
        return Mark4TrackHeader([int(w) for w in self.words[:, track]], decade=self.decade, verify=False)

    decade = None

    def __init__(self, words, ntrack=None, decade=None, ref_time=None,
                 verify=True):
        if words is None:
            words = np.zeros((5, ntrack), dtype=np.uint32)
            verify = False
            self._mutable = True
        else:
            words = np.asarray(words, dtype=np.uint32)
            self._mutable = False
        self.words = words
        if decade is not None:
            self.decade = decade
        if verify:
            self.verify()
        if decade is None and ref_time is not None:
            self.infer_decade(ref_time)

    # -- field access
    def keys(self):
        return _FIELDS.keys()

    def __getitem__(self, key):
        try:
            word, bit, nbits = _FIELDS[key][:3]
        except KeyError:
            raise KeyError("Mark4Header header does not contain {0}".format(key))
        v = (self.words[word] >> np.uint32(bit)) & np.uint32((1 << nbits) - 1)
        return v.astype(bool) if nbits == 1 else v

    def __setitem__(self, key, value):
        if not self._mutable:
            raise TypeError("header is immutable; use .copy() to get a "
                            "mutable one.")
        word, bit, nbits = _FIELDS[key][:3]
        mask = np.uint32(((1 << nbits) - 1) << bit)
        if value is True:
            value = (1 << nbits) - 1                 # all bits, for masks
        value = np.asarray(value)
        if np.any(value.astype(np.int64) & ((1 << nbits) - 1) != value):
            raise ValueError("{0} cannot be represented with {1} bits"
                             .format(value, nbits))
        value = value.astype(np.uint32)
        self.words[word] = (self.words[word] & ~mask) | ((value << np.uint32(bit)) & mask)

    def copy(self):
        new = Mark4Header(self.words.copy(), decade=self.decade, verify=False)
        new._mutable = True
        return new

    @property
    def mutable(self):
        return self._mutable

    @mutable.setter
    def mutable(self, mutable):
        self._mutable = bool(mutable)

    def verify(self):
        """mark4/header.py:171-179,334-339."""
        assert self.words.shape[0] == 5
        assert np.all(self['sync_pattern'] == 0xffffffff)
        assert np.all((self['bcd_fraction'] & 0xf) % 5 != 4)
        if self.decade is not None:
            assert (1950 < self.decade < 3000)
            assert self.decade % 10 == 0, "decade must end in zero"
        assert set(self['fan_out'].tolist()) == set(range(self.fanout))
        assert (len(set(zip(self['converter_id'].tolist(),
                            self['lsb_output'].tolist()))) == self.nchan)

    @classmethod
    def fromfile(cls, fh, ntrack, decade=None, ref_time=None, verify=True):
        """Read the 160 stream words and transpose them into per-track header
        words (mark4/header.py:425-454)."""
        dtype = np.dtype(MARK4_DTYPES[ntrack])
        header_nbytes = ntrack * 160 // 8
        s = fh.read(header_nbytes)
        if len(s) != header_nbytes:
            raise EOFError("could not read full Mark 4 Header.")
        words = stream2words(np.frombuffer(s, dtype=dtype))
        return cls(words, decade=decade, ref_time=ref_time, verify=verify)

    def tofile(self, fh):
        fh.write(words2stream(self.words).tobytes())

    _properties = ('decade', 'track_id', 'fraction', 'time', 'fanout',
                   'samples_per_frame', 'bps', 'complex_data', 'nchan',
                   'sample_shape', 'nsb', 'converters')
    _invariants = {'sync_pattern', '_1_0_1_sync'}
    _stream_invariants = _invariants | {'bcd_headstack1', 'bcd_headstack2',
                                        'track_roll_enabled',
                                        'sequence_suspended', 'system_id'}

    @classmethod
    def fromvalues(cls, ntrack, decade=None, ref_time=None, **kwargs):
        """Header from keys and properties (mark4/header.py:460-507):
        defaults for headstack and track ids follow from ``ntrack``, one
        sideband unless converters or sidebands are given; ``time``, ``bps``
        and ``fanout`` make it complete.  The CRC is recalculated."""
        if ntrack in (16, 32, 64):
            # tracks 2..33 of a headstack: every one of them (32 per
            # headstack, the second headstack for tracks 32-63), or the even
            # ones for 16 tracks
            lane = np.arange(ntrack)
            step = 2 if ntrack == 16 else 1
            kwargs.setdefault('headstack_id', lane // 32)
            kwargs.setdefault('track_id', 2 + step * (lane % 32))
        if not {'lsb_output', 'converter_id', 'converter'} & set(kwargs):
            kwargs.setdefault('nsb', 1)
        self = cls(None, ntrack=ntrack, decade=decade)
        for key, field in _FIELDS.items():
            if len(field) > 3 and not key.startswith('_1_0_1'):
                self[key] = field[3]
        verify = kwargs.pop('verify', True)
        self.update(verify=False, **kwargs)
        if self.decade is None and ref_time is not None:
            self.infer_decade(ref_time)
        if verify:
            self.verify()
        return self

    @classmethod
    def fromkeys(cls, ntrack, decade=None, ref_time=None, verify=True, **kwargs):
        """Header from a complete set of key values, i.e.,
        ``Mark4Header.fromkeys(h.ntrack, h.decade, **h) == h``
        (base/header.py:695-723)."""
        missing = [key for key in _FIELDS if key not in kwargs]
        if missing:
            raise KeyError("missing key(s) {} for Mark4Header".format(missing))
        self = cls(None, ntrack=ntrack, decade=decade)
        for key in _FIELDS:
            self[key] = kwargs.pop(key)
        if kwargs:
            raise KeyError("Mark4Header header does not contain {0}"
                           .format(sorted(kwargs)))
        if decade is None and ref_time is not None:
            self.infer_decade(ref_time)
        if verify:
            self.verify()
        return self

    def update(self, crc=None, verify=True, **kwargs):
        """Set keys first, then properties in class order, and recalculate
        the CRC unless one is passed in (mark4/header.py:509-533,
        base/header.py:753-787)."""
        if not self._mutable:
            raise TypeError("header is immutable; use .copy() to get a "
                            "mutable one.")
        if crc is not None:
            kwargs['crc'] = crc
        for key in [k for k in kwargs if k in _FIELDS]:
            self[key] = kwargs.pop(key)
        for name in self._properties:
            if name in kwargs:
                setattr(self, name, kwargs.pop(name))
        if kwargs:
            import warnings
            warnings.warn("some keywords unused in header update: {0}"
                          .format(kwargs))
        if crc is None:
            self.update_crc()
        if verify:
            self.verify()

    def invariants(self):
        """Keys of parts shared by the headers of one stream
        (mark4/header.py:144-154); the class-level set is ``_invariants``."""
        return self._stream_invariants

    def invariant_pattern(self, invariants=None, ntrack=None):
        """(pattern, mask) as stream words (mark4/header.py:345-373): on an
        instance from its own words and the stream invariants."""
        if invariants is None:
            invariants = self.invariants()
        if not invariants:
            raise ValueError("cannot create an invariant_mask without "
                             "some invariants")
        mask = type(self)(None, ntrack=self.ntrack)
        for key in invariants:
            mask[key] = True
        return words2stream(self.words), words2stream(mask.words)

    @classmethod
    def class_invariant_pattern(cls, ntrack, invariants=None):
        """The reference's ``Mark4Header.invariant_pattern(ntrack=...)`` called
        on the class: defaults of the type invariants (the sync pattern plus
        the always-zero lowest bit of 'system_id')."""
        self = cls(None, ntrack=ntrack)
        if invariants is None:
            invariants = cls._invariants
        for key in invariants:
            if len(_FIELDS[key]) < 4:
                raise ValueError('can only set as invariant a header '
                                 'part that has a default.')
            self[key] = _FIELDS[key][3]
        return self.invariant_pattern(invariants)

    def __len__(self):
        return self.ntrack

    @property
    def track_id(self):
        """Track identifiers decoded from 'bcd_track_id' (mark4/header.py:191-198)."""
        return _bcd_decode_array(self['bcd_track_id'])

    @track_id.setter
    def track_id(self, track_id):
        self['bcd_track_id'] = np.array(
            [_bcd_encode(t) for t in np.broadcast_to(track_id, (self.ntrack,))])

    def update_crc(self):
        stream = words2stream(self.words)
        stream[-12:] = crc12_stream(stream[:-12].copy())
        self.words = stream2words(stream)

    # -- geometry (mark4/header.py:540-650)
    @property
    def ntrack(self):
        return self.words.shape[1]

    @property
    def stream_dtype(self):
        return np.dtype(MARK4_DTYPES[self.ntrack])

    @property
    def nbytes(self):
        return self.ntrack * 160 // 8

    @property
    def frame_nbytes(self):
        return self.ntrack * PAYLOAD_NBITS // 8

    @property
    def payload_nbytes(self):
        return self.frame_nbytes - self.nbytes

    # The quantities below are all views of three per-track header fields --
    # 'fan_out' (which of the fanout samples a track carries), 'magnitude_bit'
    # and 'lsb_output' / 'converter_id' -- tied together by the track tables of
    # Mark 4 memo 230.3.  Results are pinned to the reference's for 48 keyword
    # combinations (tests/golden/mark4_header_cases.json; reference:
    # mark4/header.py:558-735).
    _FANOUTS = (1, 2, 4)

    @property
    def fanout(self):
        return 1 + int(self['fan_out'].max())

    @fanout.setter
    def fanout(self, fanout):
        if fanout not in self._FANOUTS:
            raise ValueError("Mark 4 data only supports fanout=1, 2, or 4, "
                             "not {0}.".format(fanout))
        # sample number of every track: tracks come in (sign, magnitude or
        # odd/even) pairs that carry the same sample, except on 16-track tapes
        # where every other track is absent
        pair = 1 if self.ntrack == 16 else 2
        self['fan_out'] = (np.arange(self.ntrack) // pair) % fanout

    @property
    def samples_per_frame(self):
        # a stream word holds `fanout` samples of every channel, and the
        # header's place on tape counts as samples too
        return (PAYLOAD_NBITS) * self.fanout

    @samples_per_frame.setter
    def samples_per_frame(self, samples_per_frame):
        allowed = [f * PAYLOAD_NBITS for f in self._FANOUTS]
        if samples_per_frame not in allowed:
            raise ValueError("header cannot store {} samples per frame. Should be one of {}."
                             .format(samples_per_frame, ', '.join(map(str, allowed))))
        self.fanout = samples_per_frame // PAYLOAD_NBITS

    @property
    def bps(self):
        return 1 + int(bool(self['magnitude_bit'].any()))

    @bps.setter
    def bps(self, bps):
        if bps not in (1, 2):
            raise ValueError("Mark 4 data can only have bps=1 or 2, "
                             "not {0}".format(bps))
        flags = np.zeros(self.ntrack, bool)
        if bps == 2:
            # last axis of the track table: (sign track, magnitude track)
            flags[self._track_assignment(self.ntrack, 2, self.fanout)[..., 1]] = True
        self['magnitude_bit'] = flags

    complex_data = fixedvalue(False)        # (Mark 4 data are always real: "'complex_data' can only be set to False.")

    @property
    def nchan(self):
        return self.ntrack // self.fanout // self.bps

    @nchan.setter
    def nchan(self, nchan):
        self.bps = self.ntrack // self.fanout // nchan

    sample_shape = property(lambda self: (self.nchan,))

    @sample_shape.setter
    def sample_shape(self, sample_shape):
        (self.nchan,) = sample_shape

    # -- per-channel quantities stored per track
    def _channel_value(self, key):
        """Value of a per-track field on the first track of every channel."""
        return self[key][self.track_assignment[0, :, 0]]

    def _set_channel_value(self, key, values, dtype):
        """Give all tracks of channel c (every fanout sample, sign and
        magnitude) the value ``values[c]``."""
        tracks = self.track_assignment                      # (fanout, nchan, bps)
        full = np.zeros(self.ntrack, dtype)
        full[tracks] = np.asarray(values, dtype).reshape(1, -1, 1)
        self[key] = full

    @property
    def nsb(self):
        lsb = self['lsb_output']
        return 2 if lsb.min() != lsb.max() else 1

    @nsb.setter
    def nsb(self, nsb):
        """1: all tracks carry the same sideband flag (set, like the reference
        does); 2: the flag alternates from track to track, which the reference
        lays out for a 32-track headstack (mark4/header.py:654-677; a 64-track
        header is two of them).  Either way the converters get their default
        numbering."""
        if nsb not in (1, 2):
            raise ValueError("number of sidebands can only be 1 or 2.")
        if nsb == 2 and self.ntrack not in (32, 64):
            raise ValueError("two sidebands can only be set for 32 or "
                             "64 tracks.")
        self['lsb_output'] = (np.arange(self.ntrack) % 2 == 1) if nsb == 2 \
            else np.ones(self.ntrack, bool)
        ids = np.arange(self.ntrack // (self.fanout * self.bps * nsb))
        if ids.size > 2:
            # default order within groups of four: 0, 2, 1, 3
            ids = ids.reshape(-1, 2, 2).swapaxes(1, 2).reshape(-1)
        self.converters = ids

    @property
    def converters(self):
        """Converter id and sideband of every channel: structured array with
        'converter' and 'lsb' entries (mark4/header.py:690-735).  Can be set
        with such an array, a dict with those keys, or just the converter ids
        (sidebands stay as they are; with two sidebands one id per sideband
        pair suffices)."""
        table = np.zeros(self.nchan, [("converter", int), ("lsb", bool)])
        table['converter'] = self._channel_value('converter_id')
        table['lsb'] = self._channel_value('lsb_output')
        return table

    @converters.setter
    def converters(self, converters):
        nchan = self.nchan
        ids = lsb = None
        if isinstance(converters, dict) or getattr(getattr(converters, 'dtype', None), 'names', None):
            ids, lsb = np.asarray(converters['converter']), np.asarray(converters['lsb'], bool)
        else:
            ids = np.array(converters)
            if ids.size * 2 == nchan and self.nsb == 2:
                # one id per pair of sidebands: hand it to both members
                now = self._channel_value('lsb_output')
                both = np.zeros(nchan, int)
                both[now] = ids
                both[~now] = ids
                ids = both
        if ids.size != nchan:
            raise ValueError("Mark 4 file with bps={0}, fanout={1} needs to define {2} converters"
                             .format(self.bps, self.fanout, nchan))
        if lsb is not None:
            self._set_channel_value('lsb_output', lsb, bool)
        self._set_channel_value('converter_id', ids, int)

    @classmethod
    def _track_assignment(cls, ntrack, bps, fanout):
        """(fanout, nchan, bps) table of track numbers.  The memo's tables are
        for a 32-track headstack; 64 tracks are two headstacks side by side
        (channels of the second follow those of the first), 16 tracks use
        every other track of one."""
        table = _TRACK_ASSIGNMENTS.get((bps, fanout))
        if table is None:
            raise ValueError("Mark 4 reader does not support bps={0}, "
                             "fanout={1}".format(bps, fanout))
        if ntrack not in (16, 32, 64):
            raise ValueError("have Mark 4 track assignments only for "
                             "ntrack=32 or 64, not {0}".format(ntrack))
        if ntrack == 16:
            return table[:, 0::2] // 2
        return table if ntrack == 32 else np.concatenate([table, 32 + table], axis=1)

    track_assignment = property(
        lambda self: self._track_assignment(self.ntrack, self.bps, self.fanout))

    def magnitude_signature(self):
        """None for the standard sign/magnitude placement, else the packed
        magnitude bits used as decoder key (mark4/payload.py:346-357)."""
        magnitude_bit = self['magnitude_bit']
        if self.bps == 1 or np.all(magnitude_bit[self.track_assignment]
                                   == [False, True]):
            return None
        return int(np.packbits(magnitude_bit).view(self.stream_dtype).item())

    # -- time (mark4/header.py:181-262)
    def infer_decade(self, ref_time):
        year = as_time(ref_time).astype('datetime64[Y]').astype(int) + 1970
        frac = ((as_time(ref_time) - np.datetime64(str(year), 'ns'))
                / np.timedelta64(365, 'D'))
        self.decade = int(np.around(year + frac - int(self['bcd_unit_year'][0]),
                                    decimals=-1))

    @property
    def fraction(self):
        ms = _bcd_decode_array(self['bcd_fraction'])
        return (ms + (ms % 5) * 0.25) / 1000.

    @fraction.setter
    def fraction(self, fraction):
        """Fractional seconds, a multiple of 1.25 ms, stored truncated to ms
        (mark4/header.py:214-221)."""
        ms = np.asarray(fraction) * 1000.
        if np.any(np.abs((ms / 1.25) - np.around(ms / 1.25)) > 1e-6):
            raise ValueError("{0} ms is not a multiple of 1.25 ms".format(ms))
        ms = np.broadcast_to(np.floor(ms + 1e-6).astype(int), (self.ntrack,))
        self['bcd_fraction'] = np.array([_bcd_encode(m) for m in ms])

    def time_quarter_ms(self, track=0):
        """Time of `track` in units of 0.25 ms since the start of its year
        (what the scan kernel works in)."""
        day = _bcd_decode_int(self['bcd_day'][track])
        hour = _bcd_decode_int(self['bcd_hour'][track])
        minute = _bcd_decode_int(self['bcd_minute'][track])
        second = _bcd_decode_int(self['bcd_second'][track])
        ms = _bcd_decode_int(self['bcd_fraction'][track])
        return (((day * 24 + hour) * 60 + minute) * 60 + second) * 4000 + 4 * ms + ms % 5

    @property
    def year(self):
        return self.decade + int(self['bcd_unit_year'][0])

    def get_time(self):
        q = self.time_quarter_ms()
        return (np.datetime64('{:04d}-01-01'.format(self.year), 'ns')
                + np.timedelta64(q * 250000 - 86400 * 10 ** 9, 'ns'))

    def set_time(self, time):
        time = as_time(time)
        year = int(time.astype('datetime64[Y]').astype(int)) + 1970
        ns = int((time - np.datetime64('{:04d}-01-01'.format(year), 'ns'))
                 / np.timedelta64(1, 'ns'))
        ms_total, rem = divmod(ns, 1000000)
        frac_ms = ms_total % 1000 + rem / 1e6
        if abs(frac_ms / 1.25 - round(frac_ms / 1.25)) > 1e-6:
            raise ValueError("{0} ms is not a multiple of 1.25 ms".format(frac_ms))
        sec_total = ms_total // 1000
        day, sec_of_day = divmod(sec_total, 86400)
        hour, rest = divmod(sec_of_day, 3600)
        minute, second = divmod(rest, 60)
        self.decade = year // 10 * 10
        n = self.ntrack
        self['bcd_unit_year'] = np.full(n, year % 10)
        self['bcd_day'] = np.full(n, _bcd_encode(day + 1))
        self['bcd_hour'] = np.full(n, _bcd_encode(hour))
        self['bcd_minute'] = np.full(n, _bcd_encode(minute))
        self['bcd_second'] = np.full(n, _bcd_encode(second))
        self['bcd_fraction'] = np.full(n, _bcd_encode(int(np.floor(frac_ms + 1e-6))))

    time = property(get_time, set_time)

    def __eq__(self, other):
        return (type(self) is type(other)
                and np.array_equal(self.words, other.words))

    def __repr__(self):
        """Every key with its per-track values, runs of one value folded and long lists
        cut to their ends, BCD / CRC / sync fields in hex (mark4/header.py:797-813)."""
        def show(key, value):
            if key.startswith(('bcd', 'crc', 'sync_pattern')):
                try:
                    return hex(int(value))
                except Exception:
                    pass
            return str(value)
        name, outs = type(self).__name__, []
        for k in self.keys():
            v = np.atleast_1d(np.asarray(self[k]))
            if len(v) == 1:
                text = show(k, v[0])
            elif np.all(v == v[0]):
                text = '[{}]*{}'.format(show(k, v[0]), v.size)
            else:
                ends = (v[0], '...', v[-1]) if len(v) > 4 else tuple(v)
                text = '[{}]'.format(', '.join(x if isinstance(x, str) else show(k, x) for x in ends))
            outs.append('{}: {}'.format(k, text))
        return "<{} {}>".format(name, (",\n  " + " " * len(name)).join(outs))


def frame_header_streams(header0, times, invalid=None, before_invalid=False):
    """(nframes, 160) stream-word headers for frames at `times`
    (datetime64[ns] array): `header0` with each frame's time code, the
    the CRC-12 of every track recomputed -- over a header that still carries the
    PREVIOUS frame's error flag, as the reference's writer computes it -- and then
    the 'communication_error' flag of every track set where `invalid` is true
    (`before_invalid`: whether the frame before the first of `times` was invalid);
    valid frames after valid ones are what ``header0.copy(); set_time(t);
    update_crc(); words2stream(words)`` gives frame by frame
    (`set_time`, `crc12_stream`, `words2stream` above), for all frames at once."""
    from ..base.utils import bcd_encode as bcd_array
    times = np.asarray(times, dtype='M8[ns]')
    n = len(times)
    ntrack = header0.ntrack
    dtype = header0.stream_dtype
    year = times.astype('M8[Y]').astype(np.int64) + 1970
    ns = (times - times.astype('M8[Y]').astype('M8[ns]')).astype(np.int64)
    ms_total, rem = np.divmod(ns, 1000000)
    frac_ms = ms_total % 1000 + rem / 1e6
    if np.any(np.abs(frac_ms / 1.25 - np.rint(frac_ms / 1.25)) > 1e-6):
        raise ValueError("times are not multiples of 1.25 ms")
    sec_total = ms_total // 1000
    day, sec_of_day = np.divmod(sec_total, 86400)
    hour, rest = np.divmod(sec_of_day, 3600)
    minute, second = np.divmod(rest, 60)
    w3 = ((year % 10) << 28) | (bcd_array(day + 1) << 16) | (bcd_array(hour) << 8) | bcd_array(minute)
    w4 = (bcd_array(second) << 24) | (bcd_array(np.floor(frac_ms + 1e-6).astype(np.int64)) << 12)
    # Only words 3 and 4 (the time code) and one flag bit differ from frame to
    # frame, and they are THE SAME IN EVERY TRACK: their stream words (track t =
    # bit t) are all ones or zero.  Words 0-2 are transposed once.
    ones = dtype.type(np.iinfo(dtype).max)
    base = words2stream(header0.words)[:96]                    # bit 31 of a word comes first
    out = np.empty((n, 160), dtype=dtype)
    out[:, :96] = base
    shifts = np.arange(31, -1, -1, dtype=np.int64)
    out[:, 96:128] = ((w3[:, None] >> shifts) & 1).astype(dtype) * ones
    out[:, 128:160] = ((w4[:, None] >> shifts) & 1).astype(dtype) * ones      # (crc bits zero for now)
    # The error flag and the CRC, as in files the reference's writer leaves behind.  Its
    # writer keeps ONE frame object: starting frame k it sets the time -- which renews the
    # CRC over the header AS IT STANDS, still carrying frame k-1's error flag -- then clears
    # the flag, and sets it again when a piece of frame k is written as invalid
    # (base/base.py:1297-1323 `_make_frame` / `write`, mark4/frame.py `valid`; neither
    # touches the CRC).  So the CRC of frame k covers the flag of frame k-1
    # (`before_invalid` for the first frame of this call), and the flag itself is frame
    # k's (recorded cases mark4:incomplete_and_headerless_streams,
    # writers:mark5b_and_mark4_invalid_stretches).  Nothing reads the CRC back
    # (SURVEY row M4-x); the point is byte identity of written files.
    pos = 32 + 31 - _FIELDS['communication_error'][1]
    inv = np.zeros(n, bool) if invalid is None else np.asarray(invalid, bool)
    crc_flag = np.concatenate(([bool(before_invalid)], inv[:-1])) if n else inv
    out[:, pos] = 0
    out[crc_flag, pos] = ones
    # CRC-12 of the first 148 stream words of every track: polynomial division
    # without initial value is linear over GF(2), so crc bit k of every track is
    # the XOR of the stream words whose unit message has that bit in its remainder
    for k, idx in enumerate(_crc12_taps()):
        out[:, 148 + k] = np.bitwise_xor.reduce(out[:, idx], axis=1)
    out[:, pos] = 0
    out[inv, pos] = ones
    return out


_CRC12_TAPS = None


def _crc12_taps():
    """For each of the 12 CRC bits: the positions i < 148 whose unit message
    e_i leaves that bit set in the remainder of the division by 0x180f (the
    bit-serial division `crc12_stream` does, run once on the identity)."""
    global _CRC12_TAPS
    if _CRC12_TAPS is None:
        taps = np.array([b == '1' for b in '{:b}'.format(0x180f)], dtype=bool)
        work = np.zeros((148, 160), dtype=bool)
        work[np.arange(148), np.arange(148)] = True
        for i in range(148):
            work[:, i:i + 13] ^= work[:, i:i + 1] & taps
        _CRC12_TAPS = [np.nonzero(work[:, 148 + k])[0] for k in range(12)]
    return _CRC12_TAPS
