"""GUPPI/PUPPI raw headers: 80-character ``KEY     = value`` cards up to
``END`` (guppi/header.py:105-143).  A minimal card reader replaces
``astropy.io.fits.Header``; the derived properties follow
guppi/header.py:216-352.  Times are ``numpy.datetime64[ns]``."""
import operator

import numpy as np

__all__ = ['GUPPIHeader']

_MJD_UNIX = 40587


def _parse_value(text):
    text = text.strip()
    if text.startswith("'"):
        end = text.index("'", 1)
        return text[1:end].rstrip()
    text = text.split('/')[0].strip()
    if text in ('T', 'F'):
        return text == 'T'
    try:
        return int(text)
    except ValueError:
        try:
            return float(text)
        except ValueError:
            return text


def _format_card(key, value):
    if isinstance(value, bool):
        v = '{:>20}'.format('T' if value else 'F')
    elif isinstance(value, str):
        v = "'{:<8}'".format(value)
    elif isinstance(value, float):
        v = '{:>20}'.format(repr(value).upper() if 'e' in repr(value) else repr(value))
    else:
        v = '{:>20}'.format(value)
    return '{:<8}= {}'.format(key, v).ljust(80)[:80]


class GUPPIHeader(dict):
    """Dictionary of header cards with the reference's derived properties."""

    _defaults = [('BACKEND', 'GUPPI'), ('BLOCSIZE', 0), ('PKTIDX', 0),
                 ('OVERLAP', 0), ('SRC_NAME', 'unset'), ('TELESCOP', 'unset'),
                 ('PKTFMT', '1SFA'), ('PKTSIZE', 8192), ('NBITS', 8),
                 ('NPOL', 1), ('OBSNCHAN', 1)]

    def __init__(self, *args, verify=True, mutable=True, **kwargs):
        super().__init__(*args, **kwargs)
        self.mutable = mutable
        if len(self) and verify:
            self.verify()

    def verify(self):
        assert all(key in self for key in ('BLOCSIZE', 'PKTIDX'))

    @classmethod
    def fromfile(cls, fh, verify=True):
        start = fh.tell()
        cards = {}
        ncards = 0
        while True:
            line = fh.read(80).decode('ascii')
            if line == '':
                raise EOFError
            ncards += 1
            if line[:3] == 'END':
                break
            if len(line) > 8 and line[8] == '=':
                cards[line[:8].strip()] = _parse_value(line[9:])
            elif line[8:9] not in ('=', ' '):
                break
        self = cls(cards, verify=False, mutable=True)
        self._ncards = ncards
        self.mutable = False
        fh.seek(start + self.nbytes)
        if verify:
            self.verify()
        return self

    @classmethod
    def fromvalues(cls, **kwargs):
        self = cls(cls._defaults, verify=False)
        props = ('bps', 'nchan', 'npol', 'payload_nbytes', 'channels_first',
                 'overlap', 'samples_per_frame', 'sample_rate', 'start_time')
        extras = [(k, kwargs.pop(k)) for k in props if k in kwargs]
        for key, value in kwargs.items():
            self[key.upper()] = value
        for key, value in extras:
            setattr(self, key, value)
        return self

    def tofile(self, fh):
        out = ''.join(_format_card(k, v) for k, v in self.items()) + 'END'.ljust(80)
        out = out.encode('ascii')
        out += (self.nbytes - len(out)) * b'\x00'
        return fh.write(out)

    def copy(self):
        new = GUPPIHeader(self, verify=False, mutable=True)
        return new

    def __setitem__(self, key, value):
        if not getattr(self, 'mutable', True):
            raise TypeError("immutable {0} does not support assignment."
                            .format(type(self).__name__))
        super().__setitem__(key.upper(), value)

    # -- sizes and shapes (guppi/header.py:216-352)
    @property
    def nbytes(self):
        # cards without '=' (HIERARCH, COMMENT, ...) still occupy 80 bytes
        ncards = getattr(self, '_ncards', None)
        nbytes = (len(self) + 1) * 80 if ncards is None else ncards * 80
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

    @property
    def _bpcs(self):
        return int(self['OBSNCHAN']) * int(self['NPOL']) * self.bps

    @property
    def sample_rate(self):
        """Complete samples per second in Hz (overlap not included)."""
        return 1. / float(self['TBIN'])

    @sample_rate.setter
    def sample_rate(self, sample_rate):
        self['TBIN'] = 1. / abs(float(sample_rate))

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
        self.payload_nbytes = (samples_per_frame * self._bpcs + 7) // 8

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

    @property
    def start_time(self):
        day = np.datetime64('1970-01-01', 'ns') + np.timedelta64(
            int(self['STT_IMJD']) - _MJD_UNIX, 'D')
        ns = int(round((float(self['STT_SMJD']) + float(self.get('STT_OFFS', 0))) * 1e9))
        return day + np.timedelta64(ns, 'ns')

    @start_time.setter
    def start_time(self, start_time):
        t = np.datetime64(start_time, 'ns')
        days = (t - np.datetime64('1970-01-01', 'ns')) // np.timedelta64(1, 'D')
        rem = t - (np.datetime64('1970-01-01', 'ns') + np.timedelta64(int(days), 'D'))
        sec = rem / np.timedelta64(1, 's')
        self['STT_IMJD'] = int(days) + _MJD_UNIX
        self['STT_SMJD'] = int(sec)
        self['STT_OFFS'] = float(sec - int(sec))

    @property
    def time(self):
        return self.start_time + np.timedelta64(int(round(self.offset * 1e9)), 'ns')
