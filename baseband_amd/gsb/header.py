"""GSB timestamp-file headers (gsb/header.py:131-361): one text line per
frame.  Rawdump lines carry the PC time only; phased lines carry PC time,
GPS time, sequence number and memory block.  Times are IST (UTC+5:30) in the
file and ``numpy.datetime64[ns]`` UTC here."""
import numpy as np
from ..base.quantities import as_time, seconds

__all__ = ['GSBHeader', 'GSBRawdumpHeader', 'GSBPhasedHeader']

_IST = np.timedelta64(330, 'm')


def _offset(utc_offset):
    """The file's clock minus UTC as a timedelta64: IST by default, or what a
    caller gives (seconds, a Quantity or timedelta; gsb/header.py:156,172-179)."""
    if utc_offset is None:
        return _IST
    if isinstance(utc_offset, np.timedelta64):
        return utc_offset.astype('timedelta64[ns]')
    return np.timedelta64(int(round(seconds(utc_offset) * 1e9)), 'ns')


def _parse_time(items, offset=_IST):
    y, mo, d, h, mi, s = (int(x) for x in items[:6])
    frac = items[6]
    ns = int(round(float(frac) * 1e9))
    base = np.datetime64('{:04d}-{:02d}-{:02d}T{:02d}:{:02d}:{:02d}'
                         .format(y, mo, d, h, mi, s), 'ns')
    return base + np.timedelta64(ns, 'ns') - offset


def _format_time(time, precision, offset=_IST):
    """UTC ``datetime64`` -> the seven items ``YYYY MM DD HH MM SS 0.fff...``
    in the file's clock (IST), rounded to `precision` decimals (gsb/header.py:19-68)."""
    t = as_time(time) + offset
    unit = 10 ** (9 - precision)
    ticks = (int(t.astype(np.int64)) + unit // 2) // unit        # nearest tick
    sec, frac = divmod(ticks, 10 ** precision)
    text = str(np.datetime64(sec, 's'))                          # YYYY-MM-DDTHH:MM:SS
    items = text[:10].split('-') + text[11:].split(':')
    return items + ['0.{:0{}d}'.format(frac, precision)]


class GSBHeader:
    """Header = the whitespace-separated items of one timestamp line
    (gsb/header.py:131-361): ``header['gps']`` (and ``'pc'``, ``'seq_nr'``,
    ``'mem_block'`` for phased data) are its keys, `time` / `gps_time` / `pc_time` its
    instants (UTC; the file's clock is IST unless `utc_offset` says otherwise).
    ``GSBHeader(words)`` gives the class of the words' mode."""
    _gps_precision, _pc_precision = 9, 6
    mode = None
    _keys = ()
    _nwords = 0

    def __new__(cls, words=None, mode=None, **kwargs):
        if cls is GSBHeader:
            if mode is None:
                if words is None:
                    raise TypeError("cannot construct an empty GSB header without knowing the mode.")
                mode = 'rawdump' if len(words) == 7 else 'phased'
            cls = {'rawdump': GSBRawdumpHeader, 'phased': GSBPhasedHeader}[mode]
        return super().__new__(cls)

    def __init__(self, words, mode=None, nbytes=None, utc_offset=None, verify=True):
        if words is None:
            self.words, self.mutable = [''] * self._nwords, True
        else:
            self.words, self.mutable = tuple(words), False
        self._nbytes = nbytes
        self._utc_offset = _offset(utc_offset)
        if verify and words is not None:
            self.verify()

    def verify(self):
        assert len(self.words) == self._nwords

    # -- mapping protocol: dict(header), **header
    def keys(self):
        return self._keys

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __contains__(self, key):
        return key in self._keys

    def _span(self, key):
        raise KeyError(key)

    def __getitem__(self, key):
        lo, hi = self._span(key)
        if hi - lo == 1:
            return int(self.words[lo])
        return ' '.join(self.words[lo:hi])

    def __setitem__(self, key, value):
        if not self.mutable:
            raise TypeError("header is immutable; use .copy() to get a mutable one.")
        lo, hi = self._span(key)
        words = list(self.words)
        if hi - lo == 1:
            words[lo] = str(int(value))
        else:
            items = str(value).split()
            if len(items) != hi - lo:
                raise ValueError("{!r} needs {} items, got {!r}".format(key, hi - lo, value))
            words[lo:hi] = items
        self.words = words

    # -- construction
    @classmethod
    def fromfile(cls, fh, verify=True, **kwargs):
        """One line of a timestamp file; `nbytes`, `utc_offset` as for the class
        (gsb/header.py:202-229)."""
        line = fh.readline()
        if isinstance(line, bytes):
            line = line.decode('ascii')
        if line.strip() == '':
            raise EOFError
        kwargs.setdefault('nbytes', len(line))      # (the line as it is in the file: trailing blank and newline)
        return cls(line.split(), verify=verify, **kwargs)

    def tofile(self, fh):
        line = ' '.join(self.words) + '\n'
        return fh.write(line if 'b' not in getattr(fh, 'mode', 't') else line.encode('ascii'))

    @classmethod
    def _mode_of(cls, mode, kwargs):
        if mode is None:
            mode = cls.mode
        if mode is None:
            if set(kwargs) & {'pc', 'pc_time', 'seq_nr', 'mem_block'}:
                return 'phased'
            raise TypeError("cannot construct a GSB header from values without knowing the mode.")
        return mode

    @classmethod
    def fromkeys(cls, mode=None, *, nbytes=None, utc_offset=None, verify=True, **kwargs):
        """Header from the raw keys: ``gps`` (and ``pc``, ``seq_nr``, ``mem_block``
        for phased data), times as their seven-item strings; all of them and no
        others (gsb/header.py:231-238)."""
        if mode is None and cls.mode is None and 'gps' in kwargs and not (set(kwargs) - {'gps'}):
            mode = 'rawdump'
        self = GSBHeader(None, mode=cls._mode_of(mode, kwargs), nbytes=nbytes, utc_offset=utc_offset) \
            if cls is GSBHeader else cls(None, nbytes=nbytes, utc_offset=utc_offset)
        if set(kwargs) != set(self._keys):
            raise KeyError("need keyword arguments for all keys in header: {}"
                           .format(sorted(set(self._keys) ^ set(kwargs))))
        for key, value in kwargs.items():
            self[key] = value
        if verify:
            self.verify()
        return self

    @classmethod
    def fromvalues(cls, mode=None, *, nbytes=None, utc_offset=None, verify=True, **kwargs):
        """Header from keys and / or `time` / `gps_time` / `pc_time` (UTC), `seq_nr`,
        `mem_block` (gsb/header.py:200-318).  The mode has to be given or to follow
        from the keywords (any of the phased ones) or the class."""
        mode = cls._mode_of(mode, kwargs)
        self = GSBHeader(None, mode=mode, nbytes=nbytes, utc_offset=utc_offset) if cls is GSBHeader \
            else cls(None, nbytes=nbytes, utc_offset=utc_offset)
        if 'time' in kwargs:
            t = kwargs.pop('time')
            kwargs.setdefault('gps_time', t)
        if mode == 'phased':
            kwargs.setdefault('seq_nr', 0)
            kwargs.setdefault('mem_block', 0)
            if 'pc' not in kwargs and 'pc_time' not in kwargs:
                if 'gps_time' in kwargs:
                    kwargs['pc_time'] = kwargs['gps_time']
        for key in [k for k in kwargs if k in self._keys]:
            self[key] = kwargs.pop(key)
        for key in ('gps_time', 'pc_time'):
            if key in kwargs:
                setattr(self, key, kwargs.pop(key))
        kwargs.pop('mode', None)
        if any(w == '' for w in self.words):
            raise TypeError("a GSB header needs a time.")
        if verify:
            self.verify()
        return self

    def update(self, *, verify=True, **kwargs):
        """New times / counters; keyword names as for `fromvalues`."""
        if 'time' in kwargs:
            t = kwargs.pop('time')
            kwargs.setdefault('gps_time', t)
            if self.mode == 'phased':
                kwargs.setdefault('pc_time', t)
        was, self.mutable = self.mutable, True
        try:
            for key in [k for k in kwargs if k in self._keys]:
                self[key] = kwargs.pop(key)
            for key in ('gps_time', 'pc_time'):
                if key in kwargs:
                    setattr(self, key, kwargs.pop(key))
        finally:
            self.mutable = was
        if verify:
            self.verify()

    def copy(self):
        new = type(self)(list(self.words), nbytes=self._nbytes, utc_offset=self._utc_offset, verify=False)
        new.words, new.mutable = list(self.words), True
        return new

    __copy__ = copy

    @property
    def utc_offset(self):
        """The file's clock minus UTC in seconds (5.5 h: GMRT writes IST)."""
        return float(self._utc_offset / np.timedelta64(1, 's'))

    @property
    def nbytes(self):
        return len(' '.join(self.words)) + 1 if self._nbytes is None else self._nbytes

    def seek_offset(self, n, nbytes=None):
        """Bytes to move a file pointer `n` timestamp lines on: rawdump lines
        all have one length; phased lines grow with the digits of the
        sequence number (gsb/header.py:240-262,319-357)."""
        if nbytes is None:
            nbytes = self.nbytes
        guess = n * nbytes
        if self.mode != 'phased':
            return guess
        seq = self['seq_nr']
        ndseq, target = len(str(seq)), seq + n
        ndtarg = len(str(target))
        while ndseq != ndtarg:
            if n > 0:
                guess += target - int('1' + ndseq * '0')
                ndseq += 1
            else:
                guess += int('1' + (ndseq - 1) * '0') - target
                ndseq -= 1
        return guess

    def __eq__(self, other):
        return type(self) is type(other) and tuple(self.words) == tuple(other.words)

    def __repr__(self):
        return "<{} {}>".format(type(self).__name__, ", ".join("{}: {}".format(k, self[k]) for k in self._keys))

    # -- times
    def _get_time(self, key, precision):
        lo, hi = self._span(key)
        return _parse_time(self.words[lo:hi], self._utc_offset)

    def _set_time(self, key, precision, time):
        self[key] = ' '.join(_format_time(time, precision, self._utc_offset))

    gps_time = property(lambda self: self._get_time('gps', self._gps_precision),
                        lambda self, t: self._set_time('gps', self._gps_precision, t))
    time = gps_time


class GSBRawdumpHeader(GSBHeader):
    mode = 'rawdump'
    _keys = ('gps',)
    _nwords = 7

    def _span(self, key):
        if key == 'gps':
            return 0, 7
        raise KeyError(key)

    pc_time = GSBHeader.gps_time


class GSBPhasedHeader(GSBHeader):
    mode = 'phased'
    _keys = ('pc', 'gps', 'seq_nr', 'mem_block')
    _nwords = 16

    def _span(self, key):
        try:
            return {'pc': (0, 7), 'gps': (7, 14), 'seq_nr': (14, 15), 'mem_block': (15, 16)}[key]
        except KeyError:
            raise KeyError(key) from None

    pc_time = property(lambda self: self._get_time('pc', self._pc_precision),
                       lambda self, t: self._set_time('pc', self._pc_precision, t))
