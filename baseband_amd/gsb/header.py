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
    """Header = the whitespace-separated items of one timestamp line."""
    _gps_precision, _pc_precision = 9, 6

    @classmethod
    def fromvalues(cls, mode=None, *, time=None, gps_time=None, pc_time=None,
                   seq_nr=0, mem_block=0, utc_offset=None, nbytes=None, **ignored):
        """Header for `time` (UTC): rawdump = the GPS time; phased = PC time,
        GPS time, sequence number and memory block (gsb/header.py:200-318)."""
        if mode is None:
            mode = 'phased' if (pc_time is not None or seq_nr or mem_block) else 'rawdump'
        gps_time = time if gps_time is None else gps_time
        if gps_time is None:
            raise TypeError("a GSB header needs a time.")
        offset = _offset(utc_offset)
        words = _format_time(gps_time, cls._gps_precision, offset)
        if mode == 'phased':
            pc = _format_time(gps_time if pc_time is None else pc_time, cls._pc_precision, offset)
            words = pc + words + [str(int(seq_nr)), str(int(mem_block))]
        return GSBHeader(words, nbytes=nbytes, utc_offset=offset)

    def tofile(self, fh):
        line = ' '.join(self.words) + '\n'
        return fh.write(line if 'b' not in getattr(fh, 'mode', 't') else line.encode('ascii'))

    def keys(self):
        return ('pc', 'gps', 'seq_nr', 'mem_block') if self.mode == 'phased' else ('gps',)

    mutable = True

    def verify(self):
        pass

    def copy(self):
        return GSBHeader(list(self.words), nbytes=self._nbytes, utc_offset=self._utc_offset)

    @classmethod
    def fromkeys(cls, mode=None, **kwargs):
        """Header from the raw keys: ``gps`` (and ``pc``, ``seq_nr``,
        ``mem_block`` for phased data), times as their seven-item strings
        (gsb/header.py:231-238)."""
        if mode is None:
            mode = 'phased' if set(kwargs) & {'pc', 'seq_nr', 'mem_block'} else 'rawdump'
        words = kwargs['gps'].split()
        if mode == 'phased':
            words = (kwargs['pc'].split() + words
                     + [str(int(kwargs.get('seq_nr', 0))), str(int(kwargs.get('mem_block', 0)))])
        return GSBHeader(words)

    def update(self, *, verify=True, **kwargs):
        """New times / counters; keyword names as for `fromvalues`."""
        current = dict(gps_time=self.time)
        if self.mode == 'phased':
            current.update(pc_time=self.pc_time, seq_nr=self['seq_nr'], mem_block=self['mem_block'])
        if 'time' in kwargs:
            t = kwargs.pop('time')
            kwargs.setdefault('gps_time', t)
            if self.mode == 'phased':
                kwargs.setdefault('pc_time', t)
        current.update(kwargs)
        current.setdefault('utc_offset', self._utc_offset)
        self.words = GSBHeader.fromvalues(self.mode, **current).words

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

    def __new__(cls, words=None, mode=None, **kwargs):
        if cls is GSBHeader and words is not None:
            cls = GSBRawdumpHeader if len(words) == 7 else GSBPhasedHeader
        return super().__new__(cls)

    def __init__(self, words, mode=None, nbytes=None, utc_offset=None, verify=True):
        self.words = tuple(words)
        self._nbytes = nbytes
        self._utc_offset = _offset(utc_offset)
        if verify:
            self.verify()

    @property
    def utc_offset(self):
        """The file's clock minus UTC in seconds (5.5 h: GMRT writes IST)."""
        return float(self._utc_offset / np.timedelta64(1, 's'))

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

    @property
    def nbytes(self):
        return len(' '.join(self.words)) + 1 if self._nbytes is None else self._nbytes

    def __eq__(self, other):
        return type(self) is type(other) and self.words == other.words


class GSBRawdumpHeader(GSBHeader):
    mode = 'rawdump'

    def verify(self):
        assert len(self.words) == 7

    @property
    def pc_time(self):
        return _parse_time(self.words[:7], self._utc_offset)

    time = gps_time = pc_time

    def __getitem__(self, key):
        if key == 'gps':
            return ' '.join(self.words[:7])
        raise KeyError(key)


class GSBPhasedHeader(GSBHeader):
    mode = 'phased'

    def verify(self):
        assert len(self.words) == 16

    def __getitem__(self, key):
        if key == 'seq_nr':
            return int(self.words[14])
        if key == 'mem_block':
            return int(self.words[15])
        if key == 'pc':
            return ' '.join(self.words[:7])
        if key == 'gps':
            return ' '.join(self.words[7:14])
        raise KeyError(key)

    @property
    def pc_time(self):
        return _parse_time(self.words[:7], self._utc_offset)

    @property
    def gps_time(self):
        return _parse_time(self.words[7:14], self._utc_offset)

    time = gps_time
