"""Mark 5B headers (host side): four words, BCD time stamp, fixed 10016-byte
frames (mark5b/header.py:60-68,91-97,177-185,235-262).  Times are
``numpy.datetime64[ns]``; ``kday`` resolves the modulo-1000 day count."""
import numpy as np
from ..base.utils import fixedvalue

from ..base.header import BitFieldHeader, four_word_struct
from ..base.quantities import as_time, hz

__all__ = ['Mark5BHeader', 'bcd_decode', 'bcd_encode', 'crc16_mark5b',
           'frame_header_words']

_MJD_UNIX = 40587          # MJD of 1970-01-01


def bcd_decode(value):
    """Binary-coded decimal -> int; ValueError on a non-decimal nibble
    (base/utils.py:18-40)."""
    result, factor = 0, 1
    while value > 0:
        value, digit = divmod(value, 16)
        if digit > 9:
            raise ValueError("invalid BCD encoded value {0}={1}."
                             .format(value, hex(value)))
        result += digit * factor
        factor *= 10
    return result


def bcd_encode(value):
    result, shift = 0, 0
    value = int(value)
    while value > 0:
        value, digit = divmod(value, 10)
        result += digit << shift
        shift += 4
    return result


def crc16_mark5b(words):
    """CRC-16-IBM, x^16 + x^15 + x^2 + 1 = 0x18005 (mark5b/header.py:20-31,154-160)
    over the 48 time-stamp bits: bcd_jday/bcd_seconds word and bcd_fraction."""
    stream = (int(words[2]) << 16) | (int(words[3]) >> 16)
    nbits, poly, ncrc = 48, 0x18005, 16
    value = stream << ncrc
    for i in range(nbits - 1, -1, -1):
        if value & (1 << (i + ncrc)):
            value ^= poly << i
    return value & 0xffff


class Mark5BHeader(BitFieldHeader):
    _fields = {
        'sync_pattern': (0, 0, 32, 0xABADDEED),
        'user': (1, 16, 16),
        'internal_tvg': (1, 15, 1),
        'frame_nr': (1, 0, 15),
        'bcd_jday': (2, 20, 12),
        'bcd_seconds': (2, 0, 20),
        'bcd_fraction': (3, 16, 16),
        'crc': (3, 0, 16),
    }
    _struct = four_word_struct
    _stream_invariants = {'sync_pattern', 'user'}
    payload_nbytes = fixedvalue(10000)
    frame_nbytes = fixedvalue(10016)
    complex_data = fixedvalue(False)
    kday = None

    def __init__(self, words=None, kday=None, ref_time=None, verify=True):
        if kday is not None:
            self.kday = kday
        super().__init__(words, verify=verify)
        if kday is None and ref_time is not None:
            self.infer_kday(ref_time)

    @classmethod
    def fromfile(cls, fh, kday=None, ref_time=None, verify=True):
        s = fh.read(16)
        if len(s) != 16:
            raise EOFError
        return cls(four_word_struct.unpack(s), kday=kday, ref_time=ref_time,
                   verify=verify)

    @classmethod
    def fromvalues(cls, *, time=None, frame_rate=None, verify=True, **kwargs):
        self = cls(None, verify=False)
        self['sync_pattern'] = None
        for key in [k for k in kwargs if k in cls._fields]:
            self[key] = kwargs.pop(key)
        for key in ('kday', 'jday', 'seconds', 'fraction'):
            if key in kwargs:
                setattr(self, key, kwargs.pop(key))
        if kwargs:
            raise KeyError("unknown header keywords: {}".format(sorted(kwargs)))
        if time is not None:
            self.set_time(time, frame_rate)
        self['crc'] = crc16_mark5b(self.words)
        if verify:
            self.verify()
        return self

    def update(self, *, time=None, frame_rate=None, crc=None, verify=True, **kwargs):
        """As the base `update`; `time` is applied last and the CRC of the time
        code is recalculated unless one is given (mark5b/header.py:127-165)."""
        super().update(verify=False, **kwargs)
        if time is not None:
            self.set_time(time, frame_rate)
        self['crc'] = crc16_mark5b(self.words) if crc is None else crc
        if verify:
            self.verify()

    def verify(self):
        assert len(self.words) == 4
        assert self['sync_pattern'] == 0xABADDEED
        assert self.kday is None or (33000 < self.kday < 400000)
        if self.kday is not None:
            assert self.kday % 1000 == 0, "kday must be thousands of MJD."

    def copy(self):
        new = super().copy()
        new.kday = self.kday
        return new

    def infer_kday(self, ref_time):
        """Thousands of MJD such that the time is within 500 days of ref_time
        (mark5b/header.py:160-175)."""
        ref_mjd = (as_time(ref_time) - np.datetime64('1970-01-01', 'ns')
                   ) / np.timedelta64(1, 'D') + _MJD_UNIX
        self.kday = int(np.around(ref_mjd - self.jday, decimals=-3))

    @property
    def jday(self):
        return bcd_decode(self['bcd_jday'])

    @jday.setter
    def jday(self, jday):
        self['bcd_jday'] = bcd_encode(jday)

    @property
    def seconds(self):
        return bcd_decode(self['bcd_seconds'])

    @seconds.setter
    def seconds(self, seconds):
        self['bcd_seconds'] = bcd_encode(seconds)

    @property
    def fraction(self):
        """Fractional second, 'unrounded' from the 0.1 ms stamp
        (mark5b/header.py:206-225)."""
        ns = bcd_decode(self['bcd_fraction']) * 100000
        return (156250 * ((ns + 156249) // 156250)) / 1e9

    @fraction.setter
    def fraction(self, fraction):
        ns = np.around(fraction * 1.e9)
        self['bcd_fraction'] = bcd_encode(int(ns / 100000))

    def get_time(self, frame_rate=None):
        # mark5b/header.py:262-303: the frame number is exact when the rate is
        # known; the BCD fraction is the fallback (and unusable when zero)
        nr = int(self['frame_nr'])
        if nr == 0:
            fraction = 0.
        elif frame_rate is not None:
            fraction = nr / hz(frame_rate)
        else:
            fraction = self.fraction
            if fraction == 0.:
                raise ValueError('the fractional second in the header is zero although '
                                 'the frame number is not: pass in a frame_rate.')
        days = self.kday + self.jday - _MJD_UNIX
        return (np.datetime64('1970-01-01', 'ns') + np.timedelta64(days, 'D')
                + np.timedelta64(self.seconds, 's')
                + np.timedelta64(int(round(fraction * 1e9)), 'ns'))

    def set_time(self, time, frame_rate=None):
        time = as_time(time)
        dt = int((time - np.datetime64('1970-01-01', 'ns')) / np.timedelta64(1, 'ns'))
        days, ns = divmod(dt, 86400 * 1000000000)
        mjd = days + _MJD_UNIX
        self.kday = (mjd // 1000) * 1000
        self.jday = mjd - self.kday
        int_sec, ns = divmod(ns, 1000000000)
        frame_nr, frac = 0, 0.
        if ns:
            if frame_rate is None:
                raise ValueError("cannot calculate frame rate. Pass it "
                                 "in explicitly.")
            frame_nr = int(round(ns * hz(frame_rate) / 1e9))
            frac = frame_nr / hz(frame_rate)
            if abs(frac - 1.) < 1e-9:
                int_sec, frame_nr, frac = int_sec + 1, 0, 0.
        self.seconds = int_sec
        self.fraction = frac
        self['frame_nr'] = frame_nr

    time = property(get_time, set_time)


def frame_header_words(start_time, frame_rate, first, count, user=0, internal_tvg=False):
    """(count, 4) uint32 header words of frames ``first .. first+count-1`` of
    a stream that starts at `start_time`: what ``Mark5BHeader.fromvalues(time=
    start + k / frame_rate, frame_rate=...)`` gives frame by frame
    (`set_time`, the fraction setter and `crc16_mark5b` above), for all frames
    at once -- the stream writer's per-frame Python loop cost 15 us a frame."""
    from ..base.utils import bcd_encode as bcd_array, CRC
    k = np.arange(first, first + count, dtype=np.int64)
    rate = hz(frame_rate)
    t0 = int((as_time(start_time) - np.datetime64('1970-01-01', 'ns'))
             / np.timedelta64(1, 'ns'))
    dt = t0 + np.rint(k * 1e9 / rate).astype(np.int64)
    days, ns = np.divmod(dt, 86400 * 1000000000)
    mjd = days + _MJD_UNIX
    jday = mjd - (mjd // 1000) * 1000
    int_sec, ns = np.divmod(ns, 1000000000)
    frame_nr = np.rint(ns * rate / 1e9).astype(np.int64)
    frac = frame_nr / rate
    wrap = (ns != 0) & (np.abs(frac - 1.) < 1e-9)
    int_sec = np.where(wrap, int_sec + 1, int_sec)
    frame_nr = np.where(wrap | (ns == 0), 0, frame_nr)
    frac = np.where(wrap | (ns == 0), 0., frac)
    bcd_fraction = bcd_array((np.around(frac * 1.e9) / 100000).astype(np.int64))
    words = np.empty((count, 4), dtype=np.uint32)
    words[:, 0] = 0xABADDEED
    words[:, 1] = (int(user) << 16) | (int(bool(internal_tvg)) << 15) | frame_nr
    words[:, 2] = (bcd_array(jday) << 20) | bcd_array(int_sec)
    stream = (words[:, 2].astype(np.uint64) << np.uint64(16)) | bcd_fraction.astype(np.uint64)
    crc = CRC(0x18005)(stream)
    words[:, 3] = (bcd_fraction.astype(np.uint64) << np.uint64(16)) | crc
    return words
