"""``info`` of GSB timestamp files and streams (gsb/file_info.py:16-184 in the
reference): the timestamp file alone says what it is and how many frames it lists --
an incomplete or unreadable last line is noted and ignored --; the stream adds the raw
files: payload size, number of raw files per polarisation, bandwidth, and whether the
raw files are as long as the timestamps say (`consistent`)."""
import warnings

import numpy as np

from ..base.info import _Snapshot, StreamReaderInfo

__all__ = ['GSBTimeStampInfo', 'GSBStreamReaderInfo', 'last_timestamp']


def last_timestamp(fh, header0):
    """(number of complete lines, last usable header, note or None) of a timestamp
    file: the position of the last line is worked out from the line length, which
    grows with the digits of the sequence number (`GSBHeader.seek_offset`); a last
    line that is too short, or that does not parse, is left out and the one before
    it is used (gsb/base.py:333-372, gsb/file_info.py:51-95).  `fh` is put back."""
    here = fh.tell()
    try:
        size = fh.seek(0, 2)
        guess = max(size // header0.nbytes, 1)
        while header0.seek_offset(guess) > size:
            guess -= 1
        while header0.seek_offset(guess) < size:
            guess += 1
        fh.seek(header0.seek_offset(guess - 1))
        line = fh.readline()
        items = line.split()
        note = None
        if len(" ".join(items)) < len(" ".join(header0.words)):
            note = 'last header is incomplete and is ignored'
        else:
            try:
                last = type(header0)(items, utc_offset=header0._utc_offset)
                last.time
            except Exception as exc:
                note = 'last header failed to read ({}) and is ignored'.format(str(exc))
        if note is not None:
            guess -= 1
            fh.seek(header0.seek_offset(guess - 1))
            last = type(header0).fromfile(fh, utc_offset=header0._utc_offset)
            last.time
        return guess, last, note, line
    finally:
        fh.seek(here)


class GSBTimeStampInfo(_Snapshot):
    """Snapshot of a timestamp file reader (gsb/file_info.py:16-95)."""
    attr_names = ('format', 'mode', 'number_of_frames', 'frame_rate', 'start_time',
                  'readable', 'missing', 'errors', 'warnings')

    def __init__(self, reader):
        super().__init__()
        self.title = 'GSBTimeStampIO information'
        self.format = self.mode = self.number_of_frames = self.frame_rate = self.start_time = None
        self.readable = None                    # (not known without the raw files)
        self.missing = {'raw': 'need raw binary files for the stream reader'}
        with reader.temporary_offset(0):
            self.header0 = self._guarded('header0', reader.read_timestamp)
        if self.header0 is None:
            return
        self.format, self.mode = 'gsb', self.header0.mode
        self.start_time = self._guarded('start_time', lambda: self.header0.time)
        self.frame_rate = self._guarded('frame_rate', reader.get_frame_rate)

        def count():
            n, _, note, _ = last_timestamp(reader.fh_raw, self.header0)
            if note:
                self.warnings['number_of_frames'] = note
            return n
        self.number_of_frames = self._guarded('number_of_frames', count)


class GSBStreamReaderInfo(StreamReaderInfo):
    """Snapshot of a GSB stream reader (gsb/file_info.py:98-184)."""
    attr_names = tuple(list(StreamReaderInfo.attr_names[:StreamReaderInfo.attr_names.index('readable')])
                       + ['bandwidth', 'n_raw', 'payload_nbytes']
                       + list(StreamReaderInfo.attr_names[StreamReaderInfo.attr_names.index('readable'):]))

    def __init__(self, stream):
        _Snapshot.__init__(self)
        self.title = 'GSBStream information'
        for name in ('start_time', 'stop_time', 'sample_rate', 'shape', 'bps', 'complex_data', 'verify'):
            setattr(self, name, self._guarded(name, lambda n=name: getattr(stream, n)))
        self.format = 'gsb'
        self.closed = stream.closed
        self.payload_nbytes = stream.payload_nbytes
        fh_raw = stream.fh_raw
        self.n_raw = len(fh_raw[0]) if isinstance(fh_raw, (list, tuple)) else 1
        self.bandwidth = None
        if self.sample_rate is not None and self.shape is not None:
            self.bandwidth = self.sample_rate * self.shape[-1] / (1 if self.complex_data else 2)
        self.file_info = None
        if not stream.closed:
            self.file_info = self._guarded('file_info', lambda: stream.fh_ts.info)
        if self.file_info is not None:
            self.file_info.missing.pop('raw', None)
            self.errors.update(self.file_info.errors)
            self.warnings.update(self.file_info.warnings)
        self.consistent = False
        self.readable = False
        if stream.closed:
            return
        # the first frame: when it cannot be had (raw files too short for the payload
        # size assumed), that is the one error and nothing else is checked
        # (gsb/file_info.py:108-116,178-184: `readable` needs `frame0`)
        try:
            here = stream.tell()
            try:
                stream.seek(0)
                stream._check_first_frame()
            finally:
                stream.seek(here)
        except Exception as exc:
            self.errors['frame0'] = exc
            return
        self.checks['decodable'] = self._decodable(stream)
        self.consistent = self.checks['consistent'] = self._consistent(stream)
        self.readable = all(bool(v) for v in self.checks.values())

    def _decodable(self, stream):
        here = stream.tell()
        try:
            stream.seek(0)
            stream.read(1)
            return True
        except Exception as exc:
            self.errors['decodable'] = exc
            return False
        finally:
            stream.seek(here)

    def _consistent(self, stream):
        """Whether timestamp and raw files are consistent in length."""
        try:
            pl_nbytes = self.payload_nbytes
            nchan = stream._unsliced_shape[-1]
            duration = float((self.stop_time - self.start_time) / np.timedelta64(1, 'ns')) * 1e-9
            expected_size = int(round(duration * self.sample_rate * nchan * self.bps
                                      * (2 if self.complex_data else 1) // (8 * self.n_raw)))
            fh_raw = stream.fh_raw
            if self.file_info is not None and self.file_info.mode == 'rawdump':
                fh_raw = [[fh_raw]]
            msg = ''
            try:
                for pair in fh_raw:
                    for fh in pair:
                        offset = fh.tell()
                        try:
                            fs = fh.seek(0, 2)
                        finally:
                            fh.seek(offset)
                        if fs % pl_nbytes != 0 and 'non-integer' not in msg:
                            msg += ('raw file contains non-integer number ({}) '
                                    'of payloads.'.format(fs / pl_nbytes))
                        if fs < expected_size:
                            emsg = 'raw file size smaller than expected.'
                            ratio = fs / expected_size
                            if len(pair) == 1 and 0.5 <= ratio < 0.6:
                                emsg = (emsg[:-1] + ' by {} factor of two. Are you missing the second raw file?'
                                        .format('a' if ratio == 0.5 else 'about a'))
                            raise EOFError(emsg)
                        if fs > expected_size and 'more bytes' not in msg:
                            msg += 'raw file contains more bytes than expected.'
            finally:
                if msg:
                    self.warnings['consistent'] = msg
            # as a final sanity check, read the final sample of the file
            here = stream.tell()
            try:
                stream.seek(-1, 2)
                stream.read(1)
            finally:
                stream.seek(here)
            return True
        except Exception as exc:
            self.errors['consistent'] = exc
            return False
