"""``fh.info``: what a file or stream is and whether it can be read.

Covers the part of the reference's info system that its readers and tests rely
on (base/file_info.py:282-571): standard attributes, ``missing`` arguments,
``checks`` / ``errors`` / ``warnings`` dictionaries, ``readable``, and the
``continuous`` check of stream readers.  Implemented as plain snapshot
objects: every attribute is evaluated once, inside a guard that files failures
under ``errors`` instead of raising.
"""
import warnings

__all__ = ['FileReaderInfo', 'StreamReaderInfo']


class _Snapshot:
    attr_names = ()
    title = 'information'

    def __init__(self):
        self.missing, self.checks, self.errors, self.warnings = {}, {}, {}, {}

    def _guarded(self, name, func, default=None):
        try:
            return func()
        except Exception as exc:
            self.errors[name] = exc
            return default

    def __call__(self):
        """Dict of the attributes that could be determined."""
        out = {}
        for name in self.attr_names:
            value = getattr(self, name, None)
            if value is not None and value != {}:
                out[name] = value
        return out

    def __bool__(self):
        return getattr(self, 'format', None) is not None

    def __repr__(self):
        if getattr(self, 'closed', False):
            return "File closed. Not parsable."
        lines = [self.title + ':']
        for name in self.attr_names:
            value = getattr(self, name, None)
            if value is None or isinstance(value, dict):
                continue
            lines.append('{} = {}'.format(name, value))
        for name in ('missing', 'checks', 'errors', 'warnings'):
            entries = getattr(self, name)
            if entries:
                pad = ' ' * (len(name) + 3)
                body = ('\n' + pad).join('{}: {}'.format(k, v) for k, v in entries.items())
                lines.append('\n{}:  {}'.format(name, body))
        if not self:
            lines.append('\nNot parsable. Wrong format?')
        return '\n'.join(lines)


class FileReaderInfo(_Snapshot):
    """Snapshot of a binary file reader (base/file_info.py:282-415).

    `needs` maps constructor arguments the reader lacks to the reason they are
    needed (e.g. Mark 5B: ``nchan``, ``kday``, ``ref_time``); `extras` are
    format-specific entries (VDIF ``edv``/``thread_ids``, Mark 4 ``ntrack`` ...)
    computed by the reader's ``_info_extras(header0)``."""
    attr_names = ('format', 'number_of_frames', 'frame_rate', 'sample_rate',
                  'samples_per_frame', 'sample_shape', 'bps', 'complex_data',
                  'start_time', 'readable', 'missing', 'checks', 'errors', 'warnings')

    def __init__(self, reader, fmt, needs=None):
        super().__init__()
        self.title = type(reader).__name__.replace('Reader', '') + ' information'
        self.format = None
        self.missing = dict(needs or {})
        if getattr(reader.fh_raw, 'closed', False):
            self.closed, self.header0, self.readable = True, None, False    # nothing can be asked of it
            return
        with reader.temporary_offset(0):
            header0 = self._guarded('header0', lambda: self._first_header(reader))
        self.header0 = header0
        if header0 is None:
            # frames can be there although no header could be made of them (Mark 5B
            # with an impossible kday: mark5b/file_info.py:70-75 decides the format
            # by `locate_frames` alone)
            find = getattr(reader, '_info_format_by_search', None)
            if find is not None:
                with reader.temporary_offset(0):
                    found = self._guarded('format', find)
                if found:
                    self.format = fmt
                    if isinstance(found, (list, tuple)):        # where the frames start is known all the same
                        self.offset0 = found[0]
                        self.attr_names = ('format', 'offset0') + type(self).attr_names[1:]
            self.readable = False
            return
        self.format = fmt
        offset0 = getattr(self, '_offset0', 0)
        extras = self._guarded('extras', lambda: reader._info_extras(header0, offset0), {}) or {}
        shape = extras.pop('_sample_shape', None)
        self.warnings.update(extras.pop('_warnings', None) or {})
        self.attr_names = (('format',) + tuple(extras) + type(self).attr_names[1:])
        for key, value in extras.items():
            setattr(self, key, value)
        # whole file / frame size, unless the reader counts differently
        # (Mark 4: first to last header, mark4/file_info.py:107-122)
        count = getattr(reader, '_info_number_of_frames',
                        lambda h, o: len(reader.image()) / h.frame_nbytes)
        self.number_of_frames = self._guarded('number_of_frames',
                                              lambda: count(header0, offset0))
        if self.number_of_frames is not None and self.number_of_frames % 1 == 0:
            self.number_of_frames = int(self.number_of_frames)
        elif self.number_of_frames is not None:
            self.warnings['number_of_frames'] = (
                'file contains non-integer number ({}) of frames'
                .format(self.number_of_frames))
            self.number_of_frames = None
        with reader.temporary_offset(0):
            self.frame_rate = self._guarded('frame_rate', reader.get_frame_rate)
        for name in ('samples_per_frame', 'sample_shape', 'bps', 'complex_data'):
            setattr(self, name, self._header_value(reader, header0, name))
        if shape is not None:
            self.sample_shape = tuple(shape)
        self.sample_rate = getattr(header0, 'sample_rate', None)
        if (self.sample_rate is None and self.frame_rate is not None
                and self.samples_per_frame is not None):
            self.sample_rate = self.frame_rate * self.samples_per_frame
        self.start_time = None
        if not any(k in self.missing for k in ('kday', 'decade')):
            self.start_time = self._guarded('start_time', lambda: self._time(header0))
        with reader.temporary_offset(0):
            self._decodable(reader)
        self.readable = bool(self.checks) and all(bool(v) for v in self.checks.values())

    def _first_header(self, reader):
        if getattr(reader, '_info_find_kwargs', None) is not None:
            header = reader.find_header(**reader._info_find_kwargs)
            self._offset0 = reader.tell()
            return header
        return reader.read_header()

    @staticmethod
    def _header_value(reader, header0, name):
        value = getattr(header0, name, None)
        nchan, bps = getattr(reader, 'nchan', None), getattr(reader, 'bps', None)
        if value is None and name == 'sample_shape' and nchan:
            value = (nchan,)
        if value is None and name == 'bps':
            value = bps
        if value is None and name == 'complex_data' and bps is not None:
            value = False
        if value is None and name == 'samples_per_frame' and nchan and bps:
            value = header0.payload_nbytes * 8 // bps // nchan
        return tuple(value) if name == 'sample_shape' and value is not None else value

    def _time(self, header0):
        if hasattr(header0, 'get_time'):
            try:
                return header0.get_time(frame_rate=self.frame_rate)
            except TypeError:
                return header0.get_time()
        return header0.time

    def _decodable(self, reader):
        """checks['decodable']: the first frame reads and its first sample
        decodes.  Skipped when arguments needed for that are missing."""
        if 'nchan' in self.missing:
            return
        try:
            if getattr(reader, '_info_find_kwargs', None) is not None:
                reader.find_header(**reader._info_find_kwargs)
            frame = reader.read_frame()
            frame[0]
            self.checks['decodable'] = True
        except Exception as exc:
            self.errors['decodable'] = exc
            self.checks['decodable'] = False


class StreamReaderInfo(_Snapshot):
    """Snapshot of a stream reader (base/file_info.py:417-571)."""
    attr_names = ('start_time', 'stop_time', 'sample_rate', 'shape', 'format',
                  'bps', 'complex_data', 'verify', 'readable', 'checks',
                  'errors', 'warnings')

    def __call__(self):
        """Dict of what could be determined, the raw file's under 'file_info'."""
        out = super().__call__()
        if getattr(self, 'file_info', None):
            out['file_info'] = self.file_info()
        return out

    def __init__(self, stream):
        super().__init__()
        self.title = type(stream).__name__.replace('Reader', '') + ' information'
        if stream.closed:
            # nothing can be asked of a closed file: every item is an error, the info is
            # false and says so (base/file_info.py:60-75,247-248 in the reference)
            self.closed, self.file_info, self.format, self.readable = True, None, None, False
            for name in ('start_time', 'stop_time', 'sample_rate', 'shape', 'bps', 'complex_data'):
                setattr(self, name, None)
                self.errors[name] = ValueError('I/O operation on closed file')
            self.verify = stream.verify
            return
        for name in ('start_time', 'stop_time', 'sample_rate', 'shape', 'bps',
                     'complex_data', 'verify'):
            setattr(self, name, self._guarded(name, lambda n=name: getattr(stream, n)))
        self.closed = stream.closed
        self.file_info = getattr(stream.fh_raw, 'info', None) if not stream.closed else None
        self.format = (self.file_info.format if self.file_info else
                       type(stream).__name__.split('Stream')[0].lower())
        if self.file_info is not None:
            self.checks.update(self.file_info.checks)
            self.errors.update(self.file_info.errors)
            self.warnings.update(self.file_info.warnings)
        if stream.closed or (self.file_info is not None and not self.file_info.readable):
            self.readable = False
            return
        self.checks['continuous'] = self._continuous(stream)
        self.readable = all(bool(v) for v in self.checks.values())

    def _continuous(self, fh):
        """Read one sample at the end; on failure bisect to the first frame
        that cannot be read (base/file_info.py:486-533).  With
        ``verify='fix'`` repairable problems come out as 'fixable gaps'."""
        here = fh.tell()
        spf = fh.samples_per_frame
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('error')
                good, bad = -1, None
                frame = (fh.shape[0] - 1) // spf
                while frame > good:
                    try:
                        fh.seek(frame * spf)
                        fh.read(1)
                    except Exception as exc:
                        if frame == good + 1:
                            msg = "While reading at {}: ".format(fh.tell())
                            if isinstance(exc, UserWarning):
                                self.warnings['continuous'] = msg + str(exc)
                                return 'fixable gaps'
                            self.errors['continuous'] = msg + repr(exc)
                            return False
                        bad = frame
                    else:
                        good = frame
                    if bad is not None:
                        frame = (bad + good + 1) // 2
            return 'no obvious gaps'
        finally:
            fh.seek(here)
