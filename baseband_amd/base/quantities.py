"""Arguments as callers of the reference pass them.

The reference's readers and writers take ``sample_rate`` as an
`astropy.units.Quantity`, ``ref_time`` / ``time`` / seek targets as an
`astropy.time.Time`, seek offsets as a `Quantity` of time or a `TimeDelta`
(/root/reference/baseband/vdif/base.py:422-454, base/base.py:876-917,
io/__init__.py:178); through the ``baseband.io`` entry points
(``format='vdif_hip'``) exactly those objects reach this package.  Inside,
rates are plain Hz (float), times ``numpy.datetime64[ns]`` (UTC), durations
seconds or ``numpy.timedelta64``.  The functions here convert at the seam by
DUCK TYPING -- astropy is not imported (it is not installed next to the GPU):

    Quantity   has ``to_value(unit)``
    Time       has ``utc`` and ``jd1`` / ``jd2`` (and usually ``datetime64``)
    TimeDelta  has ``jd1`` / ``jd2`` and ``to_value('s')`` but no ``utc``

This module imports nothing but numpy (tools/check_plugin_seam.py loads it next
to the real astropy, where torch is not installed).
"""
import datetime as _dt
import numbers

import numpy as np

__all__ = ['hz', 'seconds', 'nbytes', 'as_time', 'as_timedelta', 'is_time_like', 'is_duration_like',
           'normalize_kwargs', 'RATE_KEYS', 'TIME_KEYS', 'SIZE_KEYS', 'LeapSecondInstant', 'LEAP_SECOND_DAYS']

_UNIX_JD = 2440587.5            # JD of 1970-01-01T00:00:00


# Days (UTC) that END with an inserted second, 23:59:60 (IERS Bulletin C; none announced since 2016)
LEAP_SECOND_DAYS = frozenset(np.datetime64(d, 'D') for d in (
    '1972-06-30', '1972-12-31', '1973-12-31', '1974-12-31', '1975-12-31', '1976-12-31', '1977-12-31',
    '1978-12-31', '1979-12-31', '1981-06-30', '1982-06-30', '1983-06-30', '1985-06-30', '1987-12-31',
    '1989-12-31', '1990-12-31', '1992-06-30', '1993-06-30', '1994-06-30', '1995-12-31', '1997-06-30',
    '1998-12-31', '2005-12-31', '2008-12-31', '2012-06-30', '2015-06-30', '2016-12-31'))
_SEC = 10 ** 9


class LeapSecondInstant:
    """A UTC instant INSIDE an inserted leap second, ``23:59:60.f`` of `day`.

    ``numpy.datetime64`` counts 86400 seconds in every day and has no label for
    it; the reference's `astropy.time.Time` has
    (/root/reference/baseband/guppi/tests/test_guppi.py:191-199 sets and reads a
    header start time of 2012-06-30T23:59:60.375).  Header time properties of this
    package return -- and accept -- this small type for such instants and
    ``numpy.datetime64[ns]`` for all others; at the ``baseband.io`` seam it becomes a
    `Time` (plugin/_proxy.py).  Arithmetic is on the CONTINUOUS scale: the instant
    lies one second after ``23:59:59.f`` and ``1 - f`` seconds before the next
    midnight."""
    __slots__ = ('day', 'ns')

    def __init__(self, day, ns_into_second=0):
        self.day = np.datetime64(day, 'D')
        self.ns = int(ns_into_second)
        if not 0 <= self.ns < _SEC:
            raise ValueError("a leap second lasts one second")
        if self.day not in LEAP_SECOND_DAYS:
            raise ValueError("{} does not end with a leap second".format(self.day))

    @classmethod
    def fromisot(cls, text):
        """'2012-06-30T23:59:60[.fff...]' -> instance; None for any other text."""
        text = str(text).strip()
        date, _, clock = text.partition('T')
        if not clock.startswith('23:59:60'):
            return None
        frac = clock[8:]
        ns = int((frac[1:] + '0' * 9)[:9]) if frac.startswith('.') else 0
        return cls(date, ns)

    @property
    def before(self):
        """``23:59:59.f``, one second earlier (a datetime64[ns])."""
        return self.day.astype('datetime64[ns]') + np.timedelta64(86399 * _SEC + self.ns, 'ns')

    def __str__(self):
        return '{}T23:59:60.{:09d}'.format(self.day, self.ns)

    def __repr__(self):
        return "LeapSecondInstant('{}')".format(self)

    def __eq__(self, other):
        return isinstance(other, LeapSecondInstant) and (self.day, self.ns) == (other.day, other.ns)

    def __hash__(self):
        return hash((str(self.day), self.ns))

    def __sub__(self, other):
        one = np.timedelta64(_SEC, 'ns')
        if isinstance(other, LeapSecondInstant):
            a, b = self.before, other.before
            return a - b
        if isinstance(other, np.timedelta64):
            return self + (-other)
        other = np.datetime64(other, 'ns')
        midnight = (self.day + np.timedelta64(1, 'D')).astype('datetime64[ns]')
        # (labels from the next midnight on have this leap second behind them on numpy's scale as well)
        return self.before - other + (one if other < midnight else 0 * one)

    def __rsub__(self, other):
        return -(self - other)

    def __add__(self, delta):
        delta = as_timedelta(delta)
        ns = self.ns + int(delta.astype('timedelta64[ns]').astype(np.int64))
        if 0 <= ns < _SEC:
            return LeapSecondInstant(self.day, ns)
        midnight = (self.day + np.timedelta64(1, 'D')).astype('datetime64[ns]')
        if ns >= _SEC:
            return midnight + np.timedelta64(ns - _SEC, 'ns')
        return self.day.astype('datetime64[ns]') + np.timedelta64(86400 * _SEC + ns, 'ns')

    __radd__ = __add__


def _plain(x):
    return x is None or isinstance(x, (numbers.Number, np.number, str, bytes))


def hz(x):
    """Rate as a float in Hz: a number is taken as Hz, a Quantity converted
    (``32 * u.MHz`` -> 32e6; a unit that is not a frequency raises what
    astropy raises, UnitConversionError, a ValueError)."""
    if x is None:
        return None
    tv = getattr(x, 'to_value', None)
    if tv is not None:
        return float(tv('Hz'))
    return float(x)


def nbytes(x):
    """Size in bytes as an int (``file_size=512 * u.MiB`` -> 536870912)."""
    if x is None:
        return None
    tv = getattr(x, 'to_value', None)
    if tv is not None:
        return int(round(float(tv('byte'))))
    return int(x)


def _has(x, name):
    """hasattr that takes any exception for "no" (a TimeDelta asked for
    ``utc`` raises ScaleValueError, not AttributeError)."""
    try:
        getattr(x, name)
        return True
    except Exception:
        return False


def is_time_like(x):
    """An astropy-style Time (an absolute instant)."""
    return _has(x, 'jd1') and _has(x, 'jd2') and _has(x, 'utc')


def is_duration_like(x):
    """A TimeDelta or a Quantity (anything that converts itself to seconds)."""
    return not is_time_like(x) and hasattr(x, 'to_value')      # (a Quantity IS an ndarray subclass)


def _jd_to_ns(jd1, jd2):
    """Two-part Julian date -> integer ns since 1970-01-01, keeping the
    precision of the two doubles: whole days exactly, the rest in one
    rounding (jd2 is a day fraction of at most 0.5: 5e-17 day = 5 ps)."""
    jd1, jd2 = float(jd1), float(jd2)
    days = round(jd1 - _UNIX_JD)
    frac = (jd1 - _UNIX_JD - days) + jd2
    return int(days) * 86400 * 10 ** 9 + int(round(frac * 86400e9))


def as_time(x):
    """Instant as ``numpy.datetime64[ns]`` (UTC).  Accepts what
    ``numpy.datetime64`` accepts (ISO strings, `datetime`, datetime64) and
    Time-likes: through their own ``utc.datetime64`` when they have it (exact
    to the ns, leap seconds handled by astropy), else from ``utc.jd1/jd2``."""
    if x is None or isinstance(x, LeapSecondInstant):
        return x
    if isinstance(x, np.datetime64):
        return x.astype('datetime64[ns]')
    if is_time_like(x):
        utc = x.utc
        # an instant inside a leap second (23:59:60.f): numpy has no label for it
        try:
            parts = utc.ymdhms
            if int(parts['second']) >= 60:
                leap = LeapSecondInstant.fromisot(utc.isot if not hasattr(utc, 'precision') else
                                                  type(utc)(utc, precision=9).isot)
                if leap is not None:
                    return leap
        except (AttributeError, TypeError, KeyError, ValueError, IndexError):
            pass
        try:
            d = utc.datetime64
            if isinstance(d, np.ndarray):
                d = d[()] if d.ndim == 0 else d
            if isinstance(d, np.datetime64):
                return d.astype('datetime64[ns]')
        except Exception:
            pass
        return np.datetime64(_jd_to_ns(utc.jd1, utc.jd2), 'ns')
    if isinstance(x, _dt.datetime) and x.tzinfo is not None:
        x = x.astimezone(_dt.timezone.utc).replace(tzinfo=None)
    if isinstance(x, str) and 'T23:59:60' in x:
        leap = LeapSecondInstant.fromisot(x)
        if leap is not None:
            return leap
    if isinstance(x, str) and '-' not in x.strip()[1:] and ':' not in x:
        # ('56000': numpy.datetime64 would read the year 56000; astropy's Time refuses
        # a string that is not a date, and so do the readers' ref_time / time arguments)
        raise ValueError("time string {!r} is not a date (YYYY-MM-DD[Thh:mm:ss])".format(x))
    return np.datetime64(x, 'ns')


def seconds(x):
    """Duration as float seconds: number (seconds), ``numpy.timedelta64``,
    `datetime.timedelta`, Quantity of time, TimeDelta."""
    if x is None:
        return None
    if isinstance(x, np.timedelta64):
        return float(x / np.timedelta64(1, 'ns')) * 1e-9
    if isinstance(x, _dt.timedelta):
        return x.total_seconds()
    if is_duration_like(x):
        return float(x.to_value('s'))
    return float(x)


def as_timedelta(x):
    """Duration as ``numpy.timedelta64[ns]``; a TimeDelta through its two-part
    day count (exact to the ns), anything else through `seconds`."""
    if isinstance(x, np.timedelta64):
        return x.astype('timedelta64[ns]')
    if _has(x, 'jd1') and _has(x, 'jd2') and not _has(x, 'utc'):
        jd1, jd2 = float(x.jd1), float(x.jd2)
        days = round(jd1)
        return np.timedelta64(int(days) * 86400 * 10 ** 9 + int(round(((jd1 - days) + jd2) * 86400e9)), 'ns')
    return np.timedelta64(int(round(seconds(x) * 1e9)), 'ns')


RATE_KEYS = ('sample_rate', 'frame_rate', 'bandwidth')
TIME_KEYS = ('ref_time', 'time', 'start_time')
SIZE_KEYS = ('file_size',)
DURATION_KEYS = ('offset',)


def normalize_kwargs(kwargs):
    """The keyword arguments of an ``open()`` call with reference-typed values
    replaced by this package's plain ones (a new dict; plain values pass
    through untouched; ``header0`` and everything unknown are left alone)."""
    out = dict(kwargs)
    for k in RATE_KEYS:
        if k in out and not _plain(out[k]):
            out[k] = hz(out[k])
    for k in TIME_KEYS:
        if k in out and is_time_like(out[k]):
            out[k] = as_time(out[k])
    for k in SIZE_KEYS:
        if k in out and not _plain(out[k]):
            out[k] = nbytes(out[k])
    for k in DURATION_KEYS:
        if k in out and is_duration_like(out[k]):
            out[k] = seconds(out[k])
    return out
