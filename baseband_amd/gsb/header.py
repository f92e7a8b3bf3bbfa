"""GSB timestamp-file headers (gsb/header.py:131-361): one text line per
frame.  Rawdump lines carry the PC time only; phased lines carry PC time,
GPS time, sequence number and memory block.  Times are IST (UTC+5:30) in the
file and ``numpy.datetime64[ns]`` UTC here."""
import numpy as np

__all__ = ['GSBHeader', 'GSBRawdumpHeader', 'GSBPhasedHeader']

_IST = np.timedelta64(330, 'm')


def _parse_time(items):
    y, mo, d, h, mi, s = (int(x) for x in items[:6])
    frac = items[6]
    ns = int(round(float(frac) * 1e9))
    base = np.datetime64('{:04d}-{:02d}-{:02d}T{:02d}:{:02d}:{:02d}'
                         .format(y, mo, d, h, mi, s), 'ns')
    return base + np.timedelta64(ns, 'ns') - _IST


class GSBHeader:
    """Header = the whitespace-separated items of one timestamp line."""

    def __new__(cls, words=None, mode=None, **kwargs):
        if cls is GSBHeader and words is not None:
            cls = GSBRawdumpHeader if len(words) == 7 else GSBPhasedHeader
        return super().__new__(cls)

    def __init__(self, words, mode=None, verify=True):
        self.words = tuple(words)
        if verify:
            self.verify()

    @classmethod
    def fromfile(cls, fh, verify=True):
        line = fh.readline()
        if isinstance(line, bytes):
            line = line.decode('ascii')
        if line.strip() == '':
            raise EOFError
        return cls(line.split(), verify=verify)

    @property
    def nbytes(self):
        return len(' '.join(self.words)) + 1

    def __eq__(self, other):
        return type(self) is type(other) and self.words == other.words


class GSBRawdumpHeader(GSBHeader):
    mode = 'rawdump'

    def verify(self):
        assert len(self.words) == 7

    @property
    def pc_time(self):
        return _parse_time(self.words[:7])

    time = pc_time


class GSBPhasedHeader(GSBHeader):
    mode = 'phased'

    def verify(self):
        assert len(self.words) == 16

    def __getitem__(self, key):
        if key == 'seq_nr':
            return int(self.words[14])
        if key == 'mem_block':
            return int(self.words[15])
        raise KeyError(key)

    @property
    def pc_time(self):
        return _parse_time(self.words[:7])

    @property
    def gps_time(self):
        return _parse_time(self.words[7:14])

    time = gps_time
