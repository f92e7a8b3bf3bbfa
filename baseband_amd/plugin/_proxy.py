"""What the ``baseband.io`` entry points hand to callers of the reference.

``baseband.open(name, 'rs', format='vdif_hip', sample_rate=32*u.MHz)`` is
written by somebody who expects the REFERENCE's types back: ``fh.start_time``
an `astropy.time.Time`, ``fh.sample_rate`` a `Quantity`, ``fh.read()`` a NumPy
array (/root/reference/baseband/base/base.py:552-576, 876-969).  The readers of
this package return ``numpy.datetime64``, plain Hz and device tensors -- right
for code written against the package, wrong for a drop-in.  The modules of
``baseband_amd.plugin`` (what pyproject.toml registers under ``baseband.io``)
wrap the package's readers and writers in `ReferenceTyped`, which converts at
the seam and only there:

    start_time, stop_time, time, tell('time')   -> Time        (scale utc, ns exact)
    sample_rate                                 -> Quantity in Hz
    tell(unit), seek(Time | Quantity | TimeDelta | int)         as the reference
    read(count, out)                            -> numpy.ndarray (pinned, double-buffered
                                                   D2H: baseband_amd.asnumpy); `out=` a
                                                   NumPy array or a device tensor
    read_tensor(count)                          -> the device tensor, for callers who
                                                   want to stay on the GPU
    header0, frame.header, read_header()        -> `HeaderView`: time / get_time() a Time,
                                                   sample_rate / frame_rate Quantities,
                                                   offset in seconds
    info, info(), info.file_info                -> `InfoView`: the same conversions
    'rb': read_frame(), read_frameset()         -> `FrameView`: data and frame[item] NumPy
    'rb': get_frame_rate()                      -> Quantity in Hz
    everything else                             -> the wrapped object's; views passed back
                                                   in (header0=fh.header0) are unwrapped

astropy is imported lazily and only here; without it (the GPU box of this
repository's tests) the plain values come back.
"""
import importlib

import numpy as np

__all__ = ['ReferenceTyped', 'HeaderView', 'FrameView', 'PayloadView', 'InfoView', 'FileReaderView',
           'FileWriterView', 'make_module_api']


def _astropy():
    try:
        from astropy import units as u
        from astropy.time import Time
        return u, Time
    except Exception:           # not installed: plain values
        return None, None


def _as_Time(t):
    u, Time = _astropy()
    if Time is None or t is None:
        return t
    # Through the ISO text, which astropy reads on the UTC scale as written -- a leap second's
    # 23:59:60.f (`LeapSecondInstant`) included.  (A two-part Julian date made from POSIX-style
    # nanoseconds came out up to a second early on days that end with a leap second: astropy's
    # UTC day fraction is a fraction of that day's 86401 s.  ADVICE r5.)
    from ..base.quantities import LeapSecondInstant
    text = str(t) if isinstance(t, LeapSecondInstant) else str(np.datetime64(t, 'ns'))
    return Time(text, format='isot', scale='utc', precision=9)


def _as_rate(hz):
    u, _ = _astropy()
    return hz if u is None or hz is None else hz * u.Hz


def _as_seconds(sec):
    u, _ = _astropy()
    return sec if u is None or sec is None else sec * u.s


def _as_numpy(value):
    import torch
    if isinstance(value, torch.Tensor):
        from .. import asnumpy
        return asnumpy(value)
    return value


def _plain(value):
    """The package's own object behind a view (arguments going IN)."""
    return object.__getattribute__(value, '_wrapped') if isinstance(value, _View) else value


class _View:
    """An object of this package seen through the reference's types.  Subclasses
    name what is converted on the way out; everything else -- and every
    argument on the way in -- is the wrapped object's business."""
    _times = frozenset()        # attributes answered as Time
    _rates = frozenset()        # ... as Quantity in Hz
    _durations = frozenset()    # ... as Quantity in s
    _arrays = frozenset()       # device tensors -> NumPy arrays
    _views = {}                 # attribute -> view class (name) of the object it returns
    _returns = {}               # method -> converter (or view class name) of its result
    _item = None                # converter of ``view[item]``

    def __init__(self, wrapped):
        object.__setattr__(self, '_wrapped', wrapped)

    @staticmethod
    def _converter(how):
        return globals()[how] if isinstance(how, str) else how

    def __getattr__(self, name):
        value = getattr(object.__getattribute__(self, '_wrapped'), name)
        if value is None:
            return None
        if name in self._times:
            return _as_Time(value)
        if name in self._rates:
            return _as_rate(value)
        if name in self._durations:
            return _as_seconds(value)
        if name in self._arrays:
            return _as_numpy(value)
        if name in self._views:
            return self._converter(self._views[name])(value)
        if callable(value) and not isinstance(value, type):
            convert = self._converter(self._returns.get(name))

            def method(*args, **kwargs):
                got = value(*[_plain(a) for a in args], **{k: _plain(v) for k, v in kwargs.items()})
                return got if convert is None or got is None else convert(got)
            method.__name__, method.__doc__ = name, getattr(value, '__doc__', None)
            return method
        return value

    def __setattr__(self, name, value):
        setattr(self._wrapped, name, _plain(value))

    # (special methods are looked up on the type, not through __getattr__)
    def __getitem__(self, item):
        got = self._wrapped[item]
        return got if self._item is None else self._converter(self._item)(got)

    def __setitem__(self, item, value):
        self._wrapped[item] = _plain(value)

    def __delitem__(self, item):
        del self._wrapped[item]

    def __len__(self):
        return len(self._wrapped)

    def __iter__(self):
        return iter(self._wrapped)

    def __contains__(self, item):
        return item in self._wrapped

    def __eq__(self, other):
        return self._wrapped == _plain(other)

    def __ne__(self, other):
        return not self == other

    __hash__ = object.__hash__

    def __bool__(self):
        return bool(self._wrapped)

    def __enter__(self):
        self._wrapped.__enter__()
        return self

    def __exit__(self, *exc):
        return self._wrapped.__exit__(*exc)

    def __repr__(self):
        return "<reference-typed view of {!r}>".format(self._wrapped)


class HeaderView(_View):
    """A frame header: ``time`` / ``get_time()`` a Time, rates Quantities
    (vdif/header.py:400-481, dada/header.py, guppi/header.py in the reference)."""
    _times = frozenset(('time', 'start_time', 'ref_time'))
    _rates = frozenset(('sample_rate', 'frame_rate'))
    _durations = frozenset(('offset',))
    _returns = {'get_time': _as_Time, 'copy': 'HeaderView', 'track_header': 'HeaderView'}

    def __repr__(self):
        return repr(self._wrapped)

    def __str__(self):
        return str(self._wrapped)


class PayloadView(_View):
    _arrays = frozenset(('data',))
    _item = '_as_numpy'


class FrameView(_View):
    """A frame or frame set: ``data`` and ``frame[item]`` NumPy arrays
    (base/frame.py:160-199 in the reference), its header a `HeaderView`."""
    _times = HeaderView._times
    _rates = HeaderView._rates
    _durations = HeaderView._durations
    _arrays = frozenset(('data',))
    _views = {'header': 'HeaderView', 'header0': 'HeaderView', 'payload': 'PayloadView',
              'frames': lambda frames: [FrameView(f) for f in frames]}
    _returns = {'get_time': _as_Time}
    _item = '_as_numpy'


class InfoView(_View):
    """``fh.info``: times and rates in the reference's types, also inside the
    dictionary ``fh.info()`` gives (base/file_info.py:282-571 in the reference)."""
    _times = frozenset(('start_time', 'stop_time'))
    _rates = frozenset(('sample_rate', 'frame_rate'))
    _views = {'file_info': 'InfoView', 'header0': 'HeaderView'}

    def __call__(self):
        return {name: getattr(self, name) for name in self._wrapped()}

    def __repr__(self):
        return type(self._wrapped).__repr__(self)       # the same lines, with the converted values


class FileReaderView(_View):
    """A binary file reader ('rb'): frames, headers and rates as the reference's
    (vdif/base.py:51-214, base/base.py:154-406 there)."""
    _views = {'info': 'InfoView'}
    _returns = {'read_frame': 'FrameView', 'read_frameset': 'FrameView',
                'read_header': 'HeaderView', 'find_header': 'HeaderView',
                'get_frame_rate': _as_rate}


class FileWriterView(_View):
    """A binary file writer ('wb'): takes NumPy samples, the reference's headers
    and the views above."""


class ReferenceTyped(_View):
    """Proxy of a stream reader / writer of this package that answers with the
    reference's types (see the module docstring)."""
    _times = frozenset(('start_time', 'stop_time', 'time'))
    _rates = frozenset(('sample_rate',))
    _views = {'header0': 'HeaderView', 'info': 'InfoView', 'fh_raw': 'FileReaderView'}

    def tell(self, unit=None):
        got = self._wrapped.tell(unit)
        return _as_Time(got) if unit == 'time' else got

    def read(self, count=None, out=None):
        """Samples as a NumPy array (the reference's ``read``); ``out`` may be
        a NumPy array or a device tensor (then that is what comes back)."""
        import torch
        got = self._wrapped.read(count, out=out)
        return got if isinstance(out, torch.Tensor) else _as_numpy(got)

    def read_tensor(self, count=None, out=None):
        """The decoded samples where they are: a device tensor."""
        wrapped = self._wrapped
        was, wrapped.host_results = getattr(wrapped, 'host_results', False), False
        try:
            return wrapped.read(count, out=out)
        finally:
            wrapped.host_results = was

    def write(self, data, valid=True):
        return self._wrapped.write(_plain(data), valid=valid)


def make_module_api(fmt):
    """(open, info) of ``baseband_amd.plugin.<fmt>``."""
    module = importlib.import_module('baseband_amd.' + fmt)

    def open(name, mode='rs', **kwargs):
        opened = module.open(_plain(name), mode, **{k: _plain(v) for k, v in kwargs.items()})
        if 's' in mode or len(mode) == 1:
            if hasattr(opened, 'host_results'):
                opened.host_results = True              # read() hands out NumPy arrays (base/base.py)
            return ReferenceTyped(opened)
        return (FileWriterView if 'w' in mode else FileReaderView)(opened)      # 'rb' / 'wb', GSB 'rt' / 'wt'

    open.__doc__ = ("``baseband_amd.{0}.open`` for callers of the reference (``baseband.open(..., format='{0}_hip')``): "
                    "stream readers and writers come back as `ReferenceTyped` views.".format(fmt))

    def info(name, **kwargs):
        got = module.info(_plain(name), **{k: _plain(v) for k, v in kwargs.items()})
        return got if got is None else InfoView(got)

    info.__doc__ = module.info.__doc__
    return open, info
