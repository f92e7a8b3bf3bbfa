"""DADA headers: ASCII ``KEY VALUE`` lines in a block of HDR_SIZE bytes
(default 4096) (dada/header.py:117-200,289-440).  Times are
``numpy.datetime64[ns]``; sample rates are in Hz."""
import numpy as np
from ..base.quantities import as_time, hz, seconds

__all__ = ['DADAHeader']

_MJD_UNIX = 40587
_INT_KEYS = ('FILE_SIZE', 'FILE_NUMBER', 'HDR_SIZE', 'OBS_OFFSET',
             'OBS_OVERLAP', 'NBIT', 'NDIM', 'NPOL', 'NCHAN', 'RESOLUTION', 'DSB')
_FLOAT_KEYS = ('FREQ', 'BW', 'TSAMP')


class DADAHeader(dict):
    def __init__(self, *args, verify=True, mutable=True, **kwargs):
        text_layout = None
        if len(args) == 1 and isinstance(args[0], str):
            # the header's own text (what `repr` shows: dada/header.py:464-467)
            comments, text_layout = {}, []
            args = (self._fromlines(args[0].split('\n'), comments, text_layout),)
        super().__init__(*args, **kwargs)
        # what a file's header block looked like beyond KEY VALUE: trailing
        # comments per key, and the blank / comment-only lines between them,
        # so that a header read from file is written back unchanged
        # (dada/header.py:117-154 keeps them as `comments` and `_<line>` keys)
        self.comments = {}
        self._layout = None
        if text_layout is not None:
            self.comments, self._layout = comments, text_layout
        self.mutable = mutable
        if verify and len(self):
            self.verify()

    @classmethod
    def fromkeys(cls, *args, **kwargs):
        """A header from its keywords as they are (dada/header.py:225-236; for
        compatibility with the other header classes)."""
        if not args:
            kwargs.setdefault('HEADER', 'DADA')
        return cls(*args, **kwargs)

    def __repr__(self):
        return '{0}("""'.format(type(self).__name__) + '\n'.join(self._tolines()) + '""")'

    def verify(self):
        assert all(k in self for k in ('HDR_SIZE', 'NBIT', 'NDIM', 'NPOL', 'NCHAN'))

    @staticmethod
    def _fromlines(lines, comments=None, layout=None):
        out = {}
        for line_no, line in enumerate(lines):
            parts = line.strip().split('#')
            comment = parts[1].strip() if len(parts) > 1 and parts[1] else None
            split = parts[0].strip().split()
            if not split:
                if layout is not None:
                    layout.append((None, comment))      # blank or comment-only line
                continue
            key = split[0]
            value = split[1] if len(split) > 1 else None
            if value is not None:
                if key in _INT_KEYS:
                    value = int(value)
                elif key in _FLOAT_KEYS:
                    value = float(value)
            out[key] = value
            if layout is not None:
                layout.append((key, None))
            if comments is not None and comment is not None:
                comments[key] = comment
        return out

    def _tolines(self):
        """Lines as the reference writes them (dada/header.py:146-159):
        ``KEY VALUE[ # comment]``, ``# comment`` or an empty line."""
        layout = self._layout if self._layout is not None else []
        placed = {key for key, _ in layout if key is not None}
        layout = list(layout) + [(key, None) for key in self if key not in placed]
        lines = []
        for key, text in layout:
            if key is None:
                lines.append('' if text is None else '# {}'.format(text))
            elif key in self and self[key] is not None:
                comment = self.comments.get(key)
                lines.append('{} {}'.format(key, self[key]) if comment is None
                             else '{} {} # {}'.format(key, self[key], comment))
            elif key in self:
                comment = self.comments.get(key)
                lines.append('' if comment is None else '# {}'.format(comment))
        return lines

    @classmethod
    def fromfile(cls, fh, verify=True):
        """Lines up to '# end of header', a NUL byte, or HDR_SIZE bytes
        (dada/header.py:161-200)."""
        start = fh.tell()
        hdr_size = 4096
        lines = []
        block = fh.read(65536)
        rel = 0
        if block == b'':
            raise EOFError
        ended = False
        while (rel < hdr_size or not ended) and block[rel:rel + 1] != b'\x00':
            if rel >= hdr_size and rel >= len(block):
                break
            end = block.find(b'\n', rel)
            if end < 0:
                raise EOFError
            line = block[rel:end + 1].decode('ascii')
            rel = end + 1
            if line[0] == '#' and 'end of header' in line:
                ended = True
                break
            if line.startswith('HDR_SIZE'):
                hdr_size = int(line.split()[1])
            lines.append(line)
        if rel > hdr_size:
            # (the text runs beyond what HDR_SIZE says: read on to its end, as the
            # reference does, and say so; dada/header.py:194-196)
            import warnings
            warnings.warn("Odd, read {0} bytes while the header size is {1}".format(start + rel, hdr_size))
            fh.seek(start + rel)
        else:
            fh.seek(start + hdr_size)
        comments, layout = {}, []
        values = cls._fromlines(lines, comments, layout)
        values.setdefault('HDR_SIZE', hdr_size)
        self = cls(values, verify=verify, mutable=False)
        self.comments, self._layout = comments, layout
        return self

    # keys a new header starts from (dada/header.py:65-88)
    _defaults = (('HEADER', 'DADA'), ('HDR_VERSION', '1.0'), ('HDR_SIZE', 4096),
                 ('DADA_VERSION', '1.0'), ('OBS_ID', 'unset'), ('PRIMARY', 'unset'),
                 ('SECONDARY', 'unset'), ('FILE_NAME', 'unset'), ('FILE_NUMBER', 0),
                 ('FILE_SIZE', 0), ('OBS_OFFSET', 0), ('OBS_OVERLAP', 0),
                 ('SOURCE', 'unset'), ('TELESCOPE', 'unset'), ('INSTRUMENT', 'unset'),
                 ('RECEIVER', 'unset'), ('NBIT', 8), ('NDIM', 1), ('NPOL', 1),
                 ('NCHAN', 1), ('RESOLUTION', 1), ('DSB', 1))
    # property-like keywords, applied after plain keys, in this order
    # (dada/header.py:58-62)
    _properties = ('payload_nbytes', 'frame_nbytes', 'bps', 'complex_data',
                   'sample_shape', 'sample_rate', 'sideband', 'tsamp',
                   'samples_per_frame', 'offset', 'start_time', 'time')

    @classmethod
    def fromvalues(cls, **kwargs):
        """Header from defaults + keywords; keywords named after properties
        (``time``, ``sample_rate``, ``samples_per_frame`` ...) are applied last
        (dada/header.py:245-288)."""
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

    def tofile(self, fh):
        """ASCII lines, '# end of header', NUL fill to HDR_SIZE
        (dada/header.py:202-224)."""
        out = ''.join(line + '\n' for line in self._tolines()) + '# end of header\n'
        out = out.encode('ascii')
        if len(out) > self.nbytes:
            raise ValueError("cannot write header in allocated size of "
                             "{0}".format(self.nbytes))
        return fh.write(out + (self.nbytes - len(out)) * b'\x00')

    def move_to_end(self, key, last=True):
        """Move `key` to the end (or the start) of the header lines."""
        value = self.pop(key)
        if last:
            self[key] = value
        else:
            rest = list(self.items())
            self.clear()
            self[key] = value
            for k, v in rest:
                self[k] = v
        if self._layout is not None:
            self._layout = None

    def copy(self):
        new = DADAHeader(self, verify=False, mutable=True)
        new.comments = dict(self.comments)
        new._layout = None if self._layout is None else list(self._layout)
        return new

    __copy__ = copy             # (copy.copy(header): a mutable copy, as header.copy())

    def __setitem__(self, key, value):
        if not getattr(self, 'mutable', True):
            raise TypeError("immutable {0} does not support assignment."
                            .format(type(self).__name__))
        if isinstance(value, tuple):                # (value, comment)
            value, self.comments[key.upper()] = value
        super().__setitem__(key.upper(), value)

    # -- dada/header.py:289-382
    @property
    def nbytes(self):
        return self['HDR_SIZE']

    @property
    def payload_nbytes(self):
        return self['FILE_SIZE']

    @payload_nbytes.setter
    def payload_nbytes(self, payload_nbytes):
        self['FILE_SIZE'] = int(payload_nbytes)

    @property
    def frame_nbytes(self):
        return self.nbytes + self.payload_nbytes

    @frame_nbytes.setter
    def frame_nbytes(self, frame_nbytes):
        self.payload_nbytes = frame_nbytes - self.nbytes

    @property
    def bps(self):
        return self['NBIT']

    @bps.setter
    def bps(self, bps):
        self['NBIT'] = bps

    @property
    def complex_data(self):
        return self['NDIM'] == 2

    @complex_data.setter
    def complex_data(self, complex_data):
        self['NDIM'] = 2 if complex_data else 1

    @property
    def sample_shape(self):
        return self['NPOL'], self['NCHAN']

    @sample_shape.setter
    def sample_shape(self, sample_shape):
        self['NPOL'], self['NCHAN'] = sample_shape

    @property
    def sample_rate(self):
        """Complete samples per second in Hz (TSAMP is in microseconds)."""
        return 1e6 / self['TSAMP']

    @sample_rate.setter
    def sample_rate(self, sample_rate):
        # TSAMP in us, BW in MHz keeping the sign (sideband) it had
        # (dada/header.py:346-351)
        mhz = hz(sample_rate) / 1e6
        self['TSAMP'] = 1. / abs(mhz)
        bw = mhz * self['NCHAN'] / (1 if self.complex_data else 2)
        self['BW'] = (-1 if self.get('BW', bw) < 0 else 1) * bw

    @property
    def tsamp(self):
        return self['TSAMP']

    @tsamp.setter
    def tsamp(self, tsamp):
        self['TSAMP'] = tsamp

    @property
    def sideband(self):
        return self['BW'] > 0

    @sideband.setter
    def sideband(self, sideband):
        self['BW'] = (1 if sideband else -1) * abs(self['BW'])

    @property
    def _sample_nbits(self):
        return self.bps * (2 if self.complex_data else 1) * self['NPOL'] * self['NCHAN']

    @property
    def samples_per_frame(self):
        return (self.payload_nbytes * 8
                // (self.bps * (2 if self.complex_data else 1))
                // self['NPOL'] // self['NCHAN'])

    @samples_per_frame.setter
    def samples_per_frame(self, samples_per_frame):
        old = self.payload_nbytes
        self.payload_nbytes = (samples_per_frame * self._sample_nbits + 7) // 8
        if self.samples_per_frame != samples_per_frame:
            nearest = self.samples_per_frame
            self.payload_nbytes = old
            raise ValueError("header cannot store {} samples per frame. "
                             "Nearest is {}.".format(samples_per_frame, nearest))

    @property
    def offset(self):
        """Seconds since the start of the observation."""
        return (self['OBS_OFFSET'] * 8 // self._sample_nbits) * self['TSAMP'] * 1e-6

    @offset.setter
    def offset(self, offset):
        """`offset` in seconds (float), a numpy timedelta64, a Quantity of
        time or a TimeDelta."""
        offset = seconds(offset)
        self['OBS_OFFSET'] = (int(round(offset / (self['TSAMP'] * 1e-6)))
                              * ((self._sample_nbits + 7) // 8))

    @property
    def start_time(self):
        if 'MJD_START' in self:
            mjd_int, frac = str(self['MJD_START']).split('.')
            ns = int(round(float('.' + frac) * 86400e9))
            return (np.datetime64('1970-01-01', 'ns')
                    + np.timedelta64(int(mjd_int) - _MJD_UNIX, 'D')
                    + np.timedelta64(ns, 'ns'))
        t0 = self['UTC_START']
        return np.datetime64(t0[:10] + 'T' + t0[11:], 'ns')

    @start_time.setter
    def start_time(self, start_time):
        """UTC_START as yyyy-mm-dd-hh:mm:ss[.fffffffff] and MJD_START with 15
        decimals (dada/header.py:409-421)."""
        t = as_time(start_time)
        text = str(t).replace('T', '-')
        self['UTC_START'] = text[:-10] if text.endswith('.000000000') else text
        day = t.astype('datetime64[D]')
        mjd_int = int(day.astype(np.int64)) + _MJD_UNIX
        frac = int((t - day).astype(np.int64)) / 86400e9
        self['MJD_START'] = '{0:05d}'.format(mjd_int) + '{0:17.15f}'.format(frac)[1:]

    @property
    def time(self):
        return self.start_time + np.timedelta64(int(round(self.offset * 1e9)), 'ns')

    @time.setter
    def time(self, time):
        """Sets the start time if there is none yet, else the offset
        (dada/header.py:428-444)."""
        time = as_time(time)
        if 'MJD_START' not in self:
            self.start_time = time - np.timedelta64(int(round(self.offset * 1e9)), 'ns')
        else:
            self.offset = time - self.start_time

    def __eq__(self, other):
        keys = (set(self) | set(other)) - {'MJD_START'}
        return (all(self.get(k) == other.get(k) for k in keys)
                and float(self.get('MJD_START', 0.)) == float(other.get('MJD_START', 0.)))

    __hash__ = None
