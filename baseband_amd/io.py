"""Format-independent entry points: ``file_info`` and ``open`` with format
auto-detection (io/__init__.py:99-231, base/base.py:1440-1550).

Each format is probed by opening the file as a binary reader and asking for
its ``info``; the first format whose info is valid wins.  Keyword arguments
are sorted into those the format's readers take (``used_kwargs``) and the rest
(``irrelevant_kwargs``); arguments that contradict the file show up as
``inconsistent_kwargs``.
"""
import importlib
import inspect

import numpy as np
from .base.quantities import as_time, normalize_kwargs

__all__ = ['FORMATS', 'file_info', 'open']

FORMATS = ('dada', 'guppi', 'mark4', 'mark5b', 'vdif', 'gsb')


class NoInfo:
    """Falsy stand-in when a file could not be interpreted."""

    def __init__(self, reason=None):
        self.reason = reason
        self.format = None

    def __bool__(self):
        return False

    def __repr__(self):
        return 'NoInfo({!r})'.format(self.reason)


def _accepts(cls, kwargs):
    params = inspect.signature(cls.__init__).parameters
    return {k: v for k, v in kwargs.items() if k in params}


def _consistent(key, value, info):
    """True / False / None (cannot tell): does `value` for `key` agree with
    what the file says (base/base.py:1552-1593)?"""
    have = getattr(info, key, None)
    if have is None and getattr(info, 'file_info', None) is not None:
        have = getattr(info.file_info, key, None)
    if have is not None:
        return have == value
    if key == 'nchan' and getattr(info, 'shape', None):
        shape = tuple(info.shape[1:])
        return (bool(shape) and shape[-1] == value) or int(np.prod(shape)) == value
    start = getattr(info, 'start_time', None)
    if start is not None and key in ('ref_time', 'kday', 'decade'):
        start = np.datetime64(start, 'ns')
        if key == 'ref_time':
            return abs((as_time(value) - start) / np.timedelta64(1, 'D')) < 500
        if key == 'kday':
            mjd = int(start.astype('datetime64[D]').astype(np.int64)) + 40587
            return mjd // 1000 * 1000 == value
        return int(str(start)[:3]) * 10 == value
    return None


def _format_info(fmt, name, kwargs):
    module = importlib.import_module('baseband_amd.' + fmt)
    if fmt == 'gsb':
        try:
            if 'raw' not in kwargs:
                # the timestamp file alone: what it is, what it lists, and that the raw
                # files are needed for more (gsb/file_info.py:16-95 in the reference)
                with module.open(name, 'rt') as fh:
                    info = fh.info
                if info:
                    info.used_kwargs, info.consistent_kwargs, info.inconsistent_kwargs = {}, {}, {}
                    info.irrelevant_kwargs = dict(kwargs)
                    return info
                return NoInfo("not a gsb timestamp file: {}".format(info.errors))
            with module.open(name, 'rs', **kwargs) as fh:
                info = fh.info
            info.used_kwargs, info.consistent_kwargs, info.inconsistent_kwargs = dict(kwargs), {}, {}
            info.irrelevant_kwargs = {}
            return info
        except FileNotFoundError:
            raise
        except Exception as exc:
            return NoInfo("opening as gsb raised {!r}".format(exc))
    file_cls = module.open.classes['rb']
    stream_cls = module.open.classes['rs']
    file_kwargs = _accepts(file_cls, kwargs)
    kwargs_error = None
    try:
        with module.open(name, 'rb', **file_kwargs) as fh:
            info = fh.info
    except FileNotFoundError:
        raise
    except (TypeError, ValueError) as exc:
        # arguments of the wrong type or value: the format can still be told without
        # them, and the error is reported with the info ('kwargs', as the reference
        # files it: tests/test_file_info.py::test_info_wrong_type_args there)
        kwargs_error = exc
        try:
            with module.open(name, 'rb') as fh:
                info = fh.info
        except Exception:
            return NoInfo("opening as {} raised {!r}".format(fmt, exc))
    except Exception as exc:
        return NoInfo("opening as {} raised {!r}".format(fmt, exc))
    if not info:
        no = NoInfo("not a {} file: {}".format(fmt, info.errors))
        no.info = info                  # (what the format's reader made of the file: shown when only it was asked)
        return no
    if kwargs_error is not None:
        info.errors['kwargs'] = kwargs_error
        for key in file_kwargs:
            info.missing.pop(key, None)
        if any(k in file_kwargs for k in ('kday', 'ref_time', 'decade')):
            for key in ('kday', 'ref_time', 'decade'):
                info.missing.pop(key, None)
        info.used_kwargs = dict(file_kwargs)
        info.consistent_kwargs, info.inconsistent_kwargs = {}, {}
        info.irrelevant_kwargs = {k: v for k, v in kwargs.items() if k not in file_kwargs}
        return info
    used = dict(file_kwargs)
    rest = {k: v for k, v in kwargs.items() if k not in used}
    if not info.missing and (getattr(info, 'frame_rate', None) is not None or 'sample_rate' in kwargs):
        # (without a frame rate from the file or a sample rate from the caller there is
        # no stream to open: 'frame_rate' is in the errors already, base/base.py:1453-1461)
        stream_kwargs = _accepts(stream_cls, kwargs)
        if getattr(info, 'frame_rate', None) is not None:
            # the file says how fast it runs: a `sample_rate` from the caller is not
            # needed to open it and is CHECKED against the file's instead
            # (base/base.py:1453-1461 in the reference)
            stream_kwargs.pop('sample_rate', None)
        try:
            with module.open(name, 'rs', **stream_kwargs) as fs:
                sinfo = fs.info
        except Exception as exc:
            info.errors['stream'] = exc
        else:
            used.update(stream_kwargs)
            rest = {k: v for k, v in kwargs.items() if k not in used}
            info = sinfo
    info.used_kwargs = used
    info.consistent_kwargs, info.inconsistent_kwargs, info.irrelevant_kwargs = {}, {}, {}
    for key, value in rest.items():
        verdict = _consistent(key, value, info)
        target = (info.irrelevant_kwargs if verdict is None else
                  info.consistent_kwargs if verdict else info.inconsistent_kwargs)
        target[key] = value
    return info


def file_info(name, format=None, **kwargs):
    """Info on a baseband file of unknown format: every format in `format`
    (default: all) is tried in turn (io/__init__.py:99-176)."""
    formats = FORMATS if format is None else (format,) if isinstance(format, str) else tuple(format)
    kwargs = normalize_kwargs(kwargs)
    reasons = []
    for fmt in formats:
        info = _format_info(fmt, name, dict(kwargs))
        if info:
            return info
        reasons.append(info.reason)
    if len(formats) == 1 and getattr(info, 'info', None) is not None:
        return info.info                # one format asked for: its own (falsy) info, errors and all
    return NoInfo("{} does not seem formatted as any of {}.".format(name, set(formats)))


def open(name, mode='rs', format=None, **kwargs):
    """Open a baseband file; without `format` (or with a tuple of candidates)
    the format is determined from the file (io/__init__.py:178-231)."""
    kwargs = normalize_kwargs(kwargs)
    if format is None or isinstance(format, tuple):
        if 'w' in mode:
            raise ValueError("cannot specify multiple formats for writing.")
        info = file_info(name, format, **kwargs)
        if not info:
            raise ValueError("format of file could not be auto-determined")
        format = info.format
        if getattr(info, 'missing', None) and 's' in mode:
            raise TypeError("file format {} is missing required arguments {}."
                            .format(format, info.missing))
        if getattr(info, 'inconsistent_kwargs', None):
            raise ValueError("arguments inconsistent with this {} file were passed in: {}"
                             .format(format, info.inconsistent_kwargs))
        kwargs = dict(info.used_kwargs, **info.irrelevant_kwargs)
        if mode in ('rb', 'br'):
            kwargs = _accepts(importlib.import_module('baseband_amd.' + format)
                              .open.classes['rb'], kwargs)
    try:
        module = importlib.import_module('baseband_amd.' + format)
    except ImportError:
        raise ValueError("unknown format {!r}".format(format)) from None
    return module.open(name, mode, **kwargs)
