"""GUPPI/PUPPI raw headers: 80-character ``KEY     = value`` cards up to
``END`` (guppi/header.py:105-143).  A minimal card reader replaces
``astropy.io.fits.Header``; the derived properties follow
guppi/header.py:216-352.  Times are ``numpy.datetime64[ns]``."""
import operator

import numpy as np
from ..base.quantities import as_time, hz, seconds, LeapSecondInstant, LEAP_SECOND_DAYS

__all__ = ['GUPPIHeader']

_MJD_UNIX = 40587


def _parse_card(body):
    """(value, comment) from the part of a card after ``KEY     =``."""
    text = body.strip()
    if text.startswith("'"):
        end = 1
        while True:                             # '' inside a string is a quote
            end = text.index("'", end)
            if text[end:end + 2] == "''":
                end += 2
                continue
            break
        value = text[1:end].replace("''", "'").rstrip()
        rest = text[end + 1:]
    else:
        raw, _, rest = text.partition('/')
        raw = raw.strip()
        rest = '/' + rest if _ else ''
        if raw in ('T', 'F'):
            value = raw == 'T'
        else:
            try:
                value = int(raw)
            except ValueError:
                try:
                    value = float(raw.replace('D', 'E'))
                except ValueError:
                    value = raw
    comment = rest.partition('/')[2].strip() or None
    return value, comment


def _format_float(value):
    """FITS card text of a float as astropy (<= 5.2) writes it: 16 significant
    digits, always a '.' or an exponent, exponent at least two digits."""
    text = '{:.16G}'.format(value)
    if '.' not in text and 'E' not in text:
        text += '.0'
    elif 'E' in text:
        mantissa, exponent = text.split('E')
        sign = exponent[0] if exponent[0] in '+-' else ''
        text = '{}E{}{:02d}'.format(mantissa, sign, int(exponent.lstrip('+-')))
    return text


def _format_card(key, value, comment=None):
    if isinstance(value, (bool, np.bool_)):
        v = '{:>20}'.format('T' if value else 'F')
    elif isinstance(value, str):
        v = '{:20}'.format("'{:8}'".format(value.replace("'", "''")))
    elif isinstance(value, (float, np.floating)):
        v = '{:>20}'.format(_format_float(float(value)))
    else:
        v = '{:>20}'.format(value)
    card = '{:<8}= {}'.format(key, v)
    if comment:
        card += ' / ' + comment
    return card.ljust(80)[:80]


class _NoSeekPastEnd:
    """File-like over a bytes buffer whose ``seek`` past the end is harmless
    (`fromfile` positions itself behind the header's padding)."""

    def __init__(self, fh):
        self.fh = fh

    def read(self, n):
        return self.fh.read(n)

    def tell(self):
        return self.fh.tell()

    def seek(self, pos, whence=0):
        return self.fh.seek(pos, whence)


class GUPPIHeader(dict):
    """Dictionary of header cards with the reference's derived properties."""

    supported_formats = {'1SFA', 'SIMPLE'}      # packet formats known to read correctly (guppi/header.py:69-77)

    # guppi/header.py:56-67
    _defaults = [('BACKEND', 'GUPPI'), ('BLOCSIZE', 0), ('STT_OFFS', 0), ('PKTIDX', 0),
                 ('OVERLAP', 0), ('SRC_NAME', 'unset'), ('TELESCOP', 'unset'),
                 ('PKTFMT', '1SFA'), ('PKTSIZE', 8192), ('NBITS', 8),
                 ('NPOL', 1), ('OBSNCHAN', 1)]
    # property-like keywords, applied after plain keys, in this order
    # (guppi/header.py:50-53)
    _properties = ('payload_nbytes', 'frame_nbytes', 'bps', 'nchan', 'npol',
                   'sample_shape', 'sample_rate', 'sideband', 'overlap',
                   'samples_per_frame', 'offset', 'start_time', 'time')

    def __init__(self, *args, verify=True, mutable=True, **kwargs):
        super().__init__(*args, **kwargs)
        self.comments = {}
        self._layout = None         # card order of a header read from file,
        #                             with cards that are not KEY = value kept verbatim
        self.mutable = mutable
        if len(self) and verify:
            self.verify()

    def verify(self):
        assert all(key in self for key in ('BLOCSIZE', 'PKTIDX'))

    @classmethod
    def fromkeys(cls, *args, **kwargs):
        """A header from its keywords as they are (guppi/header.py:145-152; for
        compatibility with the other header classes)."""
        return cls(*args, **kwargs)

    @classmethod
    def fromfile(cls, fh, verify=True):
        """80-character cards up to END (guppi/header.py:105-143)."""
        start = fh.tell()
        cards, comments, layout = {}, {}, []
        while True:
            line = fh.read(80).decode('ascii')
            if line == '':
                raise EOFError
            if line[:3] == 'END':
                break
            if len(line) > 8 and line[8] == '=':
                key = line[:8].strip()
                cards[key], comment = _parse_card(line[9:])
                if comment:
                    comments[key] = comment
                layout.append((key, None))
            elif line[8:9] == ' ':
                layout.append((None, line))         # HIERARCH, COMMENT, blank ...
            else:
                break
        self = cls(cards, verify=False, mutable=True)
        self.comments, self._layout = comments, layout
        self.mutable = False
        fh.seek(start + self.nbytes)
        if verify:
            self.verify()
        return self

    @classmethod
    def fromvalues(cls, **kwargs):
        """Header from defaults + keywords; keywords named after properties
        are applied last, in `_properties` order (guppi/header.py:170-214)."""
        self = cls(cls._defaults, verify=False)
        self.update(**kwargs)
        return self

    def update(self, *, verify=True, **kwargs):
        extras = [(k, kwargs.pop(k)) for k in self._properties if k in kwargs]
        for key, value in kwargs.items():
            self[key] = value
        for key, value in extras:
            setattr(self, key, value)
        if verify:
            self.verify()

    def _cards(self):
        layout = self._layout if self._layout is not None else []
        placed = {key for key, _ in layout if key is not None}
        layout = list(layout) + [(key, None) for key in self if key not in placed]
        return [text if key is None else _format_card(key, self[key], self.comments.get(key))
                for key, text in layout if key is None or key in self]

    def tofile(self, fh):
        out = (''.join(self._cards()) + 'END'.ljust(80)).encode('ascii')
        out += (self.nbytes - len(out)) * b'\x00'
        return fh.write(out)

    # -- the part of ``astropy.io.fits.Header``'s interface that callers of the
    # reference's GUPPIHeader (a fits.Header subclass, guppi/header.py:17) use on
    # a header: cards, comments, case-blind keys, text round trip
    @property
    def cards(self):
        """[(keyword, value, comment)] in file order."""
        layout = self._layout if self._layout is not None else []
        placed = {key for key, _ in layout if key is not None}
        order = [key for key, _ in layout if key is not None and key in self] + [k for k in self if k not in placed]
        return [(key, self[key], self.comments.get(key, '') or '') for key in order]

    def tostring(self, sep='', endcard=True, padding=False):
        """The header as card text (80 characters per card)."""
        cards = self._cards() + (['END'.ljust(80)] if endcard else [])
        text = sep.join(cards)
        if padding:
            text += ' ' * (-len(text) % 2880)
        return text

    @classmethod
    def fromstring(cls, data, verify=True):
        """Header from card text (``tostring`` output, or the bytes of a file)."""
        import io
        if isinstance(data, str):
            data = data.encode('ascii')
        if b'END' + b' ' * 77 not in data:
            data = data + b'END'.ljust(80)
        self = cls.fromfile(_NoSeekPastEnd(io.BytesIO(data)), verify=verify)
        return self

    def set(self, keyword, value=None, comment=None):
        """Set a card's value and / or comment (fits.Header.set)."""
        key = keyword.upper()
        if value is not None or key not in self:
            self[key] = value
        if comment is not None:
            self.comments[key] = comment

    def append(self, card):
        key, value = card[0], card[1]
        self.set(key, value, card[2] if len(card) > 2 else None)

    def remove(self, keyword, ignore_missing=False):
        key = keyword.upper()
        if key not in self:
            if ignore_missing:
                return
            raise KeyError("Keyword '{}' not found.".format(keyword))
        del self[key]

    def index(self, keyword):
        return [k for k, _, _ in self.cards].index(keyword.upper())

    def rename_keyword(self, old, new):
        old, new = old.upper(), new.upper()
        if new in self:
            raise ValueError("keyword {} already exists".format(new))
        value, comment = self[old], self.comments.pop(old, None)
        if self._layout is not None:
            self._layout = [(new if k == old else k, t) for k, t in self._layout]
        dict.__delitem__(self, old)
        dict.__setitem__(self, new, value)
        if comment:
            self.comments[new] = comment

    def __getitem__(self, key):
        try:
            return dict.__getitem__(self, key)
        except KeyError:
            if isinstance(key, str) and key.upper() != key:
                return dict.__getitem__(self, key.upper())
            raise

    def __contains__(self, key):
        return dict.__contains__(self, key) or (isinstance(key, str) and dict.__contains__(self, key.upper()))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __delitem__(self, key):
        if not getattr(self, 'mutable', True):
            raise TypeError("immutable {0} does not support deletion.".format(type(self).__name__))
        key = key.upper()
        dict.__delitem__(self, key)
        self.comments.pop(key, None)
        if self._layout is not None:
            self._layout = [(k, t) for k, t in self._layout if k != key]

    def copy(self):
        new = GUPPIHeader(self, verify=False, mutable=True)
        new.comments = dict(self.comments)
        new._layout = None if self._layout is None else list(self._layout)
        return new

    __copy__ = copy             # (copy.copy(header): a mutable copy, as header.copy())

    def __setitem__(self, key, value):
        if not getattr(self, 'mutable', True):
            raise TypeError("immutable {0} does not support assignment."
                            .format(type(self).__name__))
        if isinstance(value, tuple) and len(value) == 2 and isinstance(value[1], str):
            # ``header[key] = value, 'comment'`` as astropy.io.fits headers take it
            value, comment = value
            self.comments[key.upper()] = comment
        super().__setitem__(key.upper(), value)

    # -- sizes and shapes (guppi/header.py:216-352)
    @property
    def nbytes(self):
        # cards without '=' (HIERARCH, COMMENT, ...) still occupy 80 bytes
        nbytes = (len(self._cards()) + 1) * 80
        if int(self.get('DIRECTIO', '0')) and nbytes % 512:
            nbytes += 512 - nbytes % 512
        return nbytes

    @property
    def payload_nbytes(self):
        return int(self['BLOCSIZE'])

    @payload_nbytes.setter
    def payload_nbytes(self, nbytes):
        self['BLOCSIZE'] = int(nbytes)

    @property
    def frame_nbytes(self):
        return self.nbytes + self.payload_nbytes

    @frame_nbytes.setter
    def frame_nbytes(self, frame_nbytes):
        self.payload_nbytes = frame_nbytes - self.nbytes

    @property
    def bps(self):
        return int(self['NBITS'])

    @bps.setter
    def bps(self, bps):
        self['NBITS'] = bps

    @property
    def complex_data(self):
        return int(self['OBSNCHAN']) != 1

    @property
    def npol(self):
        return int(self['NPOL']) // (2 if self.complex_data else 1)

    @npol.setter
    def npol(self, npol):
        self['NPOL'] = npol * (2 if self.complex_data else 1)

    @property
    def nchan(self):
        return int(self['OBSNCHAN'])

    @nchan.setter
    def nchan(self, nchan):
        self['OBSNCHAN'] = operator.index(nchan)

    @property
    def sample_shape(self):
        return self.npol, self.nchan

    @sample_shape.setter
    def sample_shape(self, sample_shape):
        # (the channels first: NPOL counts the real / imaginary components too,
        # guppi/header.py:281-286)
        self.nchan = sample_shape[1]
        self.npol = sample_shape[0]

    @property
    def _bpcs(self):
        return int(self['OBSNCHAN']) * int(self['NPOL']) * self.bps

    @property
    def sample_rate(self):
        """Complete samples per second in Hz (overlap not included)."""
        return 1. / float(self['TBIN'])

    @sample_rate.setter
    def sample_rate(self, sample_rate):
        # TBIN in s; OBSBW in MHz (guppi/header.py:303-308)
        sample_rate = hz(sample_rate)
        self['TBIN'] = 1. / abs(sample_rate)
        self['OBSBW'] = (sample_rate / 1e6 * int(self['OBSNCHAN'])
                         / (1 if self.complex_data else 2))

    @property
    def sideband(self):
        return float(self['OBSBW']) > 0

    @sideband.setter
    def sideband(self, sideband):
        self['OBSBW'] = (1 if sideband else -1) * abs(self['OBSBW'])

    @property
    def channels_first(self):
        return self['PKTFMT'] != 'SIMPLE'

    @channels_first.setter
    def channels_first(self, channels_first):
        self['PKTFMT'] = '1SFA' if bool(channels_first) else 'SIMPLE'

    @property
    def samples_per_frame(self):
        return self.payload_nbytes * 8 // self._bpcs

    @samples_per_frame.setter
    def samples_per_frame(self, samples_per_frame):
        old = self.payload_nbytes
        self.payload_nbytes = (samples_per_frame * self._bpcs + 7) // 8
        if self.samples_per_frame != samples_per_frame:
            nearest = self.samples_per_frame
            self.payload_nbytes = old
            raise ValueError("header cannot store {} samples per frame. "
                             "Nearest is {}.".format(samples_per_frame, nearest))

    @property
    def overlap(self):
        return int(self['OVERLAP'])

    @overlap.setter
    def overlap(self, overlap):
        self['OVERLAP'] = operator.index(overlap)

    @property
    def offset(self):
        """Seconds since the start of the observation."""
        return ((int(self['PKTIDX']) * int(self['PKTSIZE']) * 8 // self._bpcs)
                * float(self['TBIN']))

    @offset.setter
    def offset(self, offset):
        """`offset` in seconds (float), a numpy timedelta64, a Quantity of
        time or a TimeDelta."""
        offset = seconds(offset)
        self['PKTIDX'] = int(round(offset / float(self['TBIN']) / int(self['PKTSIZE']) * ((self._bpcs + 7) // 8)))

    @property
    def start_time(self):
        """Start of the observation: midnight of STT_IMJD plus STT_SMJD + STT_OFFS
        ELAPSED seconds (guppi/header.py:366-370 in the reference, a Time plus a
        TimeDelta).  A `numpy.datetime64[ns]` -- except inside an inserted leap
        second, which only a `LeapSecondInstant` can name: the reference stores
        2012-06-30T23:59:60.375 as the NEXT day, -1 s, 0.375 s (its setter below),
        other writers as 86400.375 s of the day itself."""
        day = np.datetime64('1970-01-01', 'D') + np.timedelta64(int(self['STT_IMJD']) - _MJD_UNIX, 'D')
        ns = int(round((float(self['STT_SMJD']) + float(self.get('STT_OFFS', 0))) * 1e9))
        if -10 ** 9 <= ns < 0 and (day - np.timedelta64(1, 'D')) in LEAP_SECOND_DAYS:
            return LeapSecondInstant(day - np.timedelta64(1, 'D'), ns + 10 ** 9)
        if 86400 * 10 ** 9 <= ns < 86401 * 10 ** 9 and day in LEAP_SECOND_DAYS:
            return LeapSecondInstant(day, ns - 86400 * 10 ** 9)
        return day.astype('datetime64[ns]') + np.timedelta64(ns, 'ns')

    @start_time.setter
    def start_time(self, start_time):
        """STT_IMJD (int), STT_SMJD and STT_OFFS (floats: whole and fractional
        seconds of the day, guppi/header.py:372-385)."""
        t = as_time(start_time)
        if isinstance(t, LeapSecondInstant):
            # (the reference's arithmetic: the day count is elapsed time over 86400 s, which
            # puts the 86401st second of a day into the next one, one second before its start)
            self['STT_IMJD'] = int((t.day + np.timedelta64(1, 'D')).astype(np.int64)) + _MJD_UNIX
            self['STT_SMJD'], self['STT_OFFS'] = -1.0, t.ns / 1e9
            return
        day = t.astype('datetime64[D]')
        seconds = int((t - day).astype(np.int64)) / 1e9
        self['STT_IMJD'] = int(day.astype(np.int64)) + _MJD_UNIX
        self['STT_SMJD'], self['STT_OFFS'] = divmod(seconds, 1)

    @property
    def time(self):
        return self.start_time + np.timedelta64(int(round(self.offset * 1e9)), 'ns')

    @time.setter
    def time(self, time):
        time = as_time(time)
        if 'STT_IMJD' not in self:
            self.start_time = time - np.timedelta64(int(round(self.offset * 1e9)), 'ns')
        else:
            self.offset = time - self.start_time
