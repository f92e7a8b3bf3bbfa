"""DADA headers: ASCII ``KEY VALUE`` lines in a block of HDR_SIZE bytes
(default 4096) (dada/header.py:117-200,289-440).  Times are
``numpy.datetime64[ns]``; sample rates are in Hz."""
import numpy as np

__all__ = ['DADAHeader']

_MJD_UNIX = 40587
_INT_KEYS = ('FILE_SIZE', 'FILE_NUMBER', 'HDR_SIZE', 'OBS_OFFSET',
             'OBS_OVERLAP', 'NBIT', 'NDIM', 'NPOL', 'NCHAN', 'RESOLUTION', 'DSB')
_FLOAT_KEYS = ('FREQ', 'BW', 'TSAMP')


class DADAHeader(dict):
    def __init__(self, *args, verify=True, mutable=True, **kwargs):
        super().__init__(*args, **kwargs)
        self.mutable = mutable
        if verify and len(self):
            self.verify()

    def verify(self):
        assert all(k in self for k in ('HDR_SIZE', 'NBIT', 'NDIM', 'NPOL', 'NCHAN'))

    @staticmethod
    def _fromlines(lines):
        out = {}
        for line_no, line in enumerate(lines):
            split = line.strip().split('#')[0].strip().split()
            if not split:
                continue
            key = split[0]
            value = split[1] if len(split) > 1 else None
            if value is not None:
                if key in _INT_KEYS:
                    value = int(value)
                elif key in _FLOAT_KEYS:
                    value = float(value)
            out[key] = value
        return out

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
        while rel < hdr_size and block[rel:rel + 1] != b'\x00':
            end = block.find(b'\n', rel)
            if end < 0:
                raise EOFError
            line = block[rel:end + 1].decode('ascii')
            rel = end + 1
            if line[0] == '#' and 'end of header' in line:
                break
            if line.startswith('HDR_SIZE'):
                hdr_size = int(line.split()[1])
            lines.append(line)
        fh.seek(start + hdr_size)
        values = cls._fromlines(lines)
        values.setdefault('HDR_SIZE', hdr_size)
        return cls(values, verify=verify, mutable=False)

    @classmethod
    def fromvalues(cls, **kwargs):
        self = cls({'HEADER': 'DADA', 'HDR_VERSION': '1.0', 'HDR_SIZE': 4096,
                    'DADA_VERSION': '1.0', 'OBS_OFFSET': 0, 'NBIT': 8,
                    'NDIM': 1, 'NPOL': 1, 'NCHAN': 1, 'FILE_SIZE': 0},
                   verify=False)
        props = ('bps', 'complex_data', 'sample_shape', 'sample_rate',
                 'samples_per_frame', 'start_time')
        extras = [(k, kwargs.pop(k)) for k in props if k in kwargs]
        for key, value in kwargs.items():
            self[key.upper()] = value
        for key, value in extras:
            setattr(self, key, value)
        return self

    def tofile(self, fh):
        text = ''.join('{} {}\n'.format(k, v) for k, v in self.items()
                       if v is not None) + '# end of header\n'
        out = text.encode('ascii')
        assert len(out) <= self.nbytes
        return fh.write(out + (self.nbytes - len(out)) * b'\x00')

    def copy(self):
        return DADAHeader(self, verify=False, mutable=True)

    def __setitem__(self, key, value):
        if not getattr(self, 'mutable', True):
            raise TypeError("immutable {0} does not support assignment."
                            .format(type(self).__name__))
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
        self['TSAMP'] = 1e6 / abs(float(sample_rate))

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
        self.payload_nbytes = (samples_per_frame * self._sample_nbits + 7) // 8

    @property
    def offset(self):
        """Seconds since the start of the observation."""
        return (self['OBS_OFFSET'] * 8 // self._sample_nbits) * self['TSAMP'] * 1e-6

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
        t = np.datetime64(start_time, 'ns')
        self['UTC_START'] = str(t.astype('datetime64[s]')).replace('T', '-')

    @property
    def time(self):
        return self.start_time + np.timedelta64(int(round(self.offset * 1e9)), 'ns')
