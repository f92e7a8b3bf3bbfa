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
           'normalize_kwargs', 'RATE_KEYS', 'TIME_KEYS', 'SIZE_KEYS']

_UNIX_JD = 2440587.5            # JD of 1970-01-01T00:00:00


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
    if x is None:
        return None
    if isinstance(x, np.datetime64):
        return x.astype('datetime64[ns]')
    if is_time_like(x):
        utc = x.utc
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
