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
    everything else                             -> the wrapped object's

astropy is imported lazily and only here; without it (the GPU box of this
repository's tests) the plain values come back.
"""
import importlib

import numpy as np

__all__ = ['ReferenceTyped', 'make_module_api']


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
    t = np.datetime64(t, 'ns')
    ns = int(t.astype(np.int64))
    days, rest = divmod(ns, 86400 * 10 ** 9)
    # two-part Julian date: whole days + the day fraction, exact to well below 1 ns
    return Time(2440587.5 + days, rest / 86400e9, format='jd', scale='utc', precision=9)


def _as_rate(hz):
    u, _ = _astropy()
    return hz if u is None or hz is None else hz * u.Hz


class ReferenceTyped:
    """Proxy of a stream reader / writer of this package that answers with the
    reference's types (see the module docstring)."""

    def __init__(self, wrapped):
        object.__setattr__(self, '_wrapped', wrapped)

    # -- delegation
    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, '_wrapped'), name)

    def __setattr__(self, name, value):
        setattr(self._wrapped, name, value)

    def __enter__(self):
        self._wrapped.__enter__()
        return self

    def __exit__(self, *exc):
        return self._wrapped.__exit__(*exc)

    def __repr__(self):
        return "<reference-typed view of {!r}>".format(self._wrapped)

    # -- converted
    @property
    def start_time(self):
        return _as_Time(self._wrapped.start_time)

    @property
    def stop_time(self):
        return _as_Time(self._wrapped.stop_time)

    @property
    def time(self):
        return _as_Time(self._wrapped.time)

    @property
    def sample_rate(self):
        return _as_rate(self._wrapped.sample_rate)

    def tell(self, unit=None):
        got = self._wrapped.tell(unit)
        return _as_Time(got) if unit == 'time' else got

    def read(self, count=None, out=None):
        """Samples as a NumPy array (the reference's ``read``); ``out`` may be
        a NumPy array or a device tensor (then that is what comes back)."""
        import torch
        got = self._wrapped.read(count, out=out)
        if isinstance(got, torch.Tensor) and not isinstance(out, torch.Tensor):
            from .. import asnumpy
            return asnumpy(got)
        return got

    def read_tensor(self, count=None, out=None):
        """The decoded samples where they are: a device tensor."""
        return self._wrapped.read(count, out=out)

    def write(self, data, valid=True):
        return self._wrapped.write(data, valid=valid)


def make_module_api(fmt):
    """(open, info) of ``baseband_amd.plugin.<fmt>``."""
    module = importlib.import_module('baseband_amd.' + fmt)

    def open(name, mode='rs', **kwargs):
        opened = module.open(name, mode, **kwargs)
        streamish = hasattr(opened, 'sample_rate') and (hasattr(opened, 'read') or hasattr(opened, 'write')) \
            and hasattr(opened, 'tell') and 's' in (mode if len(mode) > 1 else mode + 's')
        return ReferenceTyped(opened) if streamish else opened

    open.__doc__ = ("``baseband_amd.{0}.open`` for callers of the reference (``baseband.open(..., format='{0}_hip')``): "
                    "stream readers and writers come back as `ReferenceTyped` views.".format(fmt))

    def info(name, **kwargs):
        return module.info(name, **kwargs)

    info.__doc__ = module.info.__doc__
    return open, info
