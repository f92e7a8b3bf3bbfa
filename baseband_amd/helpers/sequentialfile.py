"""Several files addressed as one contiguous byte stream.

Observations are recorded as long runs of files; the reader has to treat them
as one stream.  This module gives the same public surface as the reference's
``baseband.helpers.sequentialfile`` (helpers/sequentialfile.py:17-425:
`FileNameSequencer`, `SequentialFileReader`, `SequentialFileWriter`, `open`)
but is organised for the GPU staging path:

* the reader keeps a table of cumulative file sizes, filled by ``stat`` (no
  file is opened just to learn its size);
* `SequentialFileReader.host_image()` returns a `SequenceImage`: every file
  memory-mapped once, addressable by *global* byte offset, which
  `staging.WindowPipeline` / `staging.upload` copy piecewise into pinned
  buffers -- so frame windows that straddle a file boundary go to HBM without
  an intermediate concatenation.
"""
import bisect
import io
import mmap
import os
import string

import numpy as np

__all__ = ['FileNameSequencer', 'SequenceImage', 'SequentialFileBase', 'SequentialFileReader',
           'SequentialFileWriter', 'open']


class FileNameSequencer:
    """List-like source of file names made from a template.

    ``template.format(**items, file_nr=i)`` with the items taken from `header`
    (helpers/sequentialfile.py:17-87).  ``len()`` counts how many consecutive
    names, starting at 0, exist on disk.

    >>> FileNameSequencer('a{file_nr:03d}.vdif')[10]
    'a010.vdif'
    """
    _counter_keys = ('file_nr',)

    def __init__(self, template, header={}):
        self.template = self._normalize(template)
        self.items = {}
        for _, field, _, _ in string.Formatter().parse(self.template):
            if field and field not in self._counter_keys:
                self.items[field] = header[field]

    @staticmethod
    def _normalize(template):
        return template

    def _values(self, file_nr):
        """Mapping used to format name number `file_nr`."""
        values = dict(self.items)
        values.update((key, file_nr) for key in self._counter_keys)
        return values

    def __getitem__(self, file_nr):
        if file_nr < 0:
            file_nr += len(self)
            if file_nr < 0:
                raise IndexError('file number out of range.')
        return self.template.format(**self._values(file_nr))

    def __len__(self):
        n = 0
        while os.path.isfile(self[n]):
            n += 1
        return n

    def __repr__(self):
        return "{}({!r})".format(type(self).__name__, self.template)


class UpperCaseSequencer(FileNameSequencer):
    """Sequencer for formats whose header keys are upper case (DADA, GUPPI):
    template fields are case-insensitive (dada/base.py:71-96,
    guppi/base.py:62-85)."""
    _counter_keys = ('FILE_NR', 'FRAME_NR')

    @staticmethod
    def _normalize(template):
        out = []
        for text, field, spec, conv in string.Formatter().parse(template):
            out.append(text.replace('{', '{{').replace('}', '}}'))
            if field is not None:
                out.append('{' + field.upper() + ('!' + conv if conv else '')
                           + (':' + spec if spec else '') + '}')
        return ''.join(out)


def _as_name(entry):
    return os.fspath(entry) if isinstance(entry, (str, os.PathLike)) else None


class SequenceImage:
    """Read-only global byte addressing over memory-mapped files.

    Behaves like the uint8 array `staging.host_image` returns for one file:
    ``len()``, slicing (a view when the slice lies in one file, a gathered
    copy otherwise) -- plus `pieces`, which the staging code uses to copy a
    range file by file, and `header_words`, the strided header gather."""

    def __init__(self, names):
        self._maps, self._mms, starts, total = [], [], [0], 0
        for name in names:
            size = os.path.getsize(name)
            if size:
                with io.open(name, 'rb') as f:
                    mm = mmap.mmap(f.fileno(), size, access=mmap.ACCESS_READ)
                self._maps.append(np.frombuffer(mm, dtype=np.uint8))
                self._mms.append(mm)
            else:
                self._maps.append(np.empty(0, np.uint8))
                self._mms.append(None)
            total += size
            starts.append(total)
        self._starts = starts
        self.dtype = np.dtype(np.uint8)

    def __len__(self):
        return self._starts[-1]

    shape = property(lambda self: (len(self),))

    def retire(self):
        """The reader is done with the image: large mappings are torn down on
        the background thread of `staging.retire_image`."""
        from ..staging import retire_mapping
        for mm, arr in zip(self._mms, self._maps):
            if mm is not None:
                retire_mapping(mm, arr)

    def pieces(self, lo, hi):
        """Array views that together hold bytes [lo, hi)."""
        lo, hi = max(0, lo), min(hi, len(self))
        out = []
        k = max(0, bisect.bisect_right(self._starts, lo) - 1)
        while lo < hi and k < len(self._maps):
            base, end = self._starts[k], self._starts[k + 1]
            stop = min(hi, end)
            if stop > lo:
                out.append(self._maps[k][lo - base:stop - base])
                lo = stop
            k += 1
        return out

    def __getitem__(self, item):
        if not isinstance(item, slice):
            i = item + len(self) if item < 0 else item
            got = self.pieces(i, i + 1)
            if not got:
                raise IndexError(item)
            return got[0][0]
        lo, hi, step = item.indices(len(self))
        if step != 1:
            raise IndexError("only contiguous slices of a file sequence")
        got = self.pieces(lo, hi)
        if len(got) == 1:
            return got[0]
        return np.concatenate(got) if got else np.empty(0, np.uint8)

    def __array__(self, dtype=None, copy=None):
        whole = self[:]
        return whole if dtype in (None, whole.dtype) else whole.astype(dtype)

    def header_words(self, frame_nbytes, nwords, offset=0):
        """(nframes, nwords) little-endian uint32 header words of frames that
        start every `frame_nbytes` bytes from `offset`; frames whose header is
        cut off by the end of the sequence are left out.  The table is lazy:
        rows are gathered from the mappings when they are indexed (probing the
        first frames of a multi-GiB sequence must not page all of it in)."""
        return _HeaderTable(self, frame_nbytes, nwords, offset)

    def _gather_headers(self, frame_nbytes, nwords, offset, first_row, last_row):
        """Rows [first_row, last_row) of the header table as an array."""
        hb = 4 * nwords
        out = np.empty((max(last_row - first_row, 0), nwords), '<u4')
        for k, m in enumerate(self._maps):
            base, end = self._starts[k], self._starts[k + 1]
            # frames whose header lies wholly inside file k
            first = max(first_row, -(-(base - offset) // frame_nbytes))
            last = min(last_row, (end - hb - offset) // frame_nbytes + 1) if end - hb >= offset else 0
            if last > first:
                o = offset + first * frame_nbytes - base
                span = m[o:o + (last - first - 1) * frame_nbytes + hb]
                rows = np.lib.stride_tricks.as_strided(
                    span, shape=(last - first, hb), strides=(frame_nbytes, 1), writeable=False)
                out[first - first_row:last - first_row] = np.ascontiguousarray(rows).view('<u4')
            # a header straddling the end of file k
            f = (end - offset) // frame_nbytes if end > offset else -1
            if first_row <= f < last_row:
                o = offset + f * frame_nbytes
                if o < end < o + hb:
                    out[f - first_row] = self[o:o + hb].view('<u4')
        return out


class _HeaderTable:
    """Lazy (nframes, nwords) uint32 view of the headers in a `SequenceImage`:
    ``len()``, ``table[k]``, ``table[a:b]``, ``table[a:b, col]`` and
    ``numpy.asarray(table)`` gather just the rows involved."""

    def __init__(self, image, frame_nbytes, nwords, offset):
        self._image, self._fn, self._nw, self._off = image, frame_nbytes, nwords, offset
        n = len(image) - offset
        self._n = (n - 4 * nwords) // frame_nbytes + 1 if n >= 4 * nwords else 0
        self.shape = (self._n, nwords)
        self.dtype = np.dtype('<u4')

    def __len__(self):
        return self._n

    def _rows(self, lo, hi):
        return self._image._gather_headers(self._fn, self._nw, self._off, lo, hi)

    def __getitem__(self, item):
        rest = ()
        if isinstance(item, tuple):
            item, rest = item[0], item[1:]
        if isinstance(item, slice):
            lo, hi, step = item.indices(self._n)
            rows = self._rows(lo, max(hi, lo)) if step > 0 else self._rows(0, self._n)[item]
            if step > 1:
                rows = rows[::step]
        else:
            k = int(item)
            if k < 0:
                k += self._n
            if not 0 <= k < self._n:
                raise IndexError(item)
            rows = self._rows(k, k + 1)[0]
        return rows[(slice(None),) + rest] if rest and isinstance(item, slice) else (
            rows[rest] if rest else rows)

    def __array__(self, dtype=None, copy=None):
        whole = self._rows(0, self._n)
        return whole if dtype in (None, whole.dtype) else whole.astype(dtype)


class _SequentialBase:
    """Position bookkeeping shared by reader and writer."""

    def __init__(self, files, mode, opener):
        self.files = files
        self.mode = mode
        self.opener = io.open if opener is None else opener
        self._starts = [0]              # cumulative sizes of the files known so far
        self.fh = None
        self.file_nr = None
        self._closed = False

    closed = property(lambda self: self._closed)

    def _switch(self, file_nr):
        if file_nr == self.file_nr:
            return
        try:
            name = self.files[file_nr]
        except IndexError:
            raise OSError('ran out of files.') from None
        fh = name if hasattr(name, 'seek') else self.opener(name, mode=self.mode)
        if self.fh is not None:
            self.fh.close()
        self.fh, self.file_nr = fh, file_nr

    def __getattr__(self, attr):
        # anything else (name, readline, ...) comes from the file that is open
        if attr.startswith('_') or attr in ('fh', 'files'):
            raise AttributeError(attr)
        return getattr(self.fh, attr)

    def close(self):
        if self.fh is not None:
            self.fh.close()
        self._closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __repr__(self):
        current = None if self.file_nr is None else self.files[self.file_nr]
        return ("{}(files={}, mode='{}')\n# At offset: {}; open file: {!r}."
                .format(type(self).__name__, self.files, self.mode,
                        None if self.closed else self.tell(), current))


SequentialFileBase = _SequentialBase      # the reference's name for the shared base


class SequentialFileReader(_SequentialBase):
    """Read several files as if they were one (helpers/sequentialfile.py:
    199-322).  `files` is a list/tuple of names, or anything indexable by file
    number that raises `IndexError` past the end (e.g. `FileNameSequencer`)."""

    def __init__(self, files, mode='rb', opener=None):
        super().__init__(files, mode, opener)
        self._complete = False          # True once the size table covers all files
        self._pos = 0
        self._switch(0)
        self._know(1)

    # -- size table
    def _size_of(self, file_nr):
        """Size of file `file_nr`, or None when there is no such file."""
        try:
            entry = self.files[file_nr]
        except IndexError:
            return None
        name = _as_name(entry)
        if name is not None and self.opener is io.open:
            return os.path.getsize(name) if os.path.isfile(name) else None
        if file_nr == self.file_nr:
            fh, close = self.fh, False
        else:
            try:
                fh, close = (entry, False) if hasattr(entry, 'seek') else (
                    self.opener(entry, mode=self.mode), True)
            except Exception:
                return None
        here = fh.tell()
        size = fh.seek(0, 2)
        fh.seek(here)
        if close:
            fh.close()
        return size

    def _know(self, nfiles=None):
        """Extend the size table to `nfiles` files (all of them if None)."""
        while not self._complete and (nfiles is None or len(self._starts) - 1 < nfiles):
            size = self._size_of(len(self._starts) - 1)
            if size is None:
                self._complete = True
            else:
                self._starts.append(self._starts[-1] + size)

    def _locate(self, offset):
        """(file number, offset inside it) for a global offset; positions at or
        past the end map to the end of the last file."""
        while not self._complete and offset >= self._starts[-1]:
            self._know(len(self._starts))
        k = bisect.bisect_right(self._starts, offset) - 1
        k = min(k, len(self._starts) - 2)
        return k, offset - self._starts[k]

    @property
    def file_size(self):
        """Size of the file that is currently open."""
        return self._size_of(self.file_nr)

    @property
    def size(self):
        """Size of all files together."""
        self._know()
        return self._starts[-1]

    # -- file interface
    def tell(self):
        return self._pos

    def seek(self, offset, whence=0):
        if self.closed:
            raise ValueError('seek of closed file.')
        if whence == 1:
            offset += self._pos
        elif whence == 2:
            offset += self.size
        elif whence != 0:
            raise ValueError("invalid 'whence'; should be 0, 1, or 2.")
        if offset < 0:
            raise OSError('invalid offset')
        self._pos = offset
        return offset

    def _sync(self):
        """Open the file holding the current position and move there."""
        k, inner = self._locate(self._pos)
        self._switch(k)
        self.fh.seek(inner)

    def read(self, count=None):
        if self.closed:
            raise ValueError('read of closed file.')
        if count is None or count < 0:
            count = max(self.size - self._pos, 0)
        chunks = []
        while count > 0:
            self._sync()
            got = self.fh.read(count)
            if not got:
                break
            chunks.append(got)
            self._pos += len(got)
            count -= len(got)
        return chunks[0] if len(chunks) == 1 else b''.join(chunks)

    def readinto(self, buffer):
        view = memoryview(buffer).cast('B')
        data = self.read(len(view))
        view[:len(data)] = data
        return len(data)

    def readable(self):
        return True

    def seekable(self):
        return True

    def readline(self, *args):
        """One line of the file holding the current position (lines do not
        continue across files)."""
        self._sync()
        line = self.fh.readline(*args)
        self._pos += len(line)
        return line

    def memmap(self, dtype=np.uint8, mode=None, offset=None, shape=None, order='C'):
        """Map part of ONE underlying file, starting at `offset` (default: the
        current position); the position moves past the mapped bytes."""
        if self.closed:
            raise ValueError('memmap of closed file.')
        dtype = np.dtype(dtype)
        if offset is not None:
            self.seek(offset)
        k, inner = self._locate(self._pos)
        if inner == self._starts[k + 1] - self._starts[k] and self._pos < self.size:
            k, inner = k + 1, 0
        if shape is None:
            count = self.size - self._pos
            if count % dtype.itemsize:
                raise ValueError("size of available data is not a "
                                 "multiple of the data-type size.")
            shape = (count // dtype.itemsize,)
        elif not isinstance(shape, tuple):
            shape = (shape,)
        count = dtype.itemsize * int(np.prod(shape, dtype=np.int64))
        if inner + count > self._starts[k + 1] - self._starts[k]:
            raise ValueError('mmap length exceeds individual file size')
        self._switch(k)
        mm = np.memmap(self.fh, dtype, (mode or self.mode).replace('b', ''), inner, shape, order)
        self._pos += count
        return mm

    def host_image(self):
        """`SequenceImage` over all files (named files only)."""
        self._know()
        names = [_as_name(self.files[k]) for k in range(len(self._starts) - 1)]
        if any(n is None for n in names) or self.opener is not io.open:
            return np.frombuffer(self._read_all(), dtype=np.uint8)
        return SequenceImage(names)

    def _read_all(self):
        here = self._pos
        self.seek(0)
        data = self.read()
        self.seek(here)
        return data

    # -- pickling: names and position travel, open files do not
    def __getstate__(self):
        state = self.__dict__.copy()
        fh = state.pop('fh')
        if not isinstance(fh, io.IOBase):
            state['fh'] = fh            # a custom object has to pickle itself
        state['file_nr'] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault('fh', None)
        if not self._closed and self.fh is None:
            self._sync()


class SequentialFileWriter(_SequentialBase):
    """Write a byte stream into files of at most `file_size` bytes each
    (helpers/sequentialfile.py:325-378).  Not seekable."""

    def __init__(self, files, mode='w+b', file_size=None, opener=None):
        super().__init__(files, mode, opener)
        self.file_size = file_size
        self._switch(0)

    def tell(self):
        return self._starts[self.file_nr] + self.fh.tell()

    def close(self):
        if self._pw_fds:
            for fd in self._pw_fds.values():
                os.close(fd)
            self._pw_fds = {}
        super().close()

    def _advance(self):
        self._starts.append(self._starts[-1] + self.fh.tell())
        self._switch(self.file_nr + 1)

    def write(self, data):
        if self.closed:
            raise ValueError('write to closed file.')
        data = memoryview(data).cast('B')
        written = 0
        while self.file_size is not None and len(data) - written > self.file_size - self.fh.tell():
            room = self.file_size - self.fh.tell()
            self.fh.write(data[written:written + room])
            written += room
            self._advance()
        self.fh.write(data[written:])
        return len(data)

    def writable(self):
        return True

    # -- positional writes (round 5): the stream writers' background sink fills
    # SEVERAL files of a sequence at the same time -- one new file takes 11-12 GB/s
    # through the page cache whatever the number of threads, 2 / 4 / 8 files at once
    # 23 / 40 / 69 GB/s (profiles/r05g_exp_file_write2.log)
    @property
    def can_pwrite(self):
        """Positional writes need files of a fixed size, named on disk, opened the plain way."""
        if self.file_size is None or self.opener is not io.open or self.closed:
            return False
        try:
            return _as_name(self.files[self.file_nr]) is not None
        except Exception:
            return False

    def pwrite_stream(self, data, offset):
        """`data` at byte `offset` of the STREAM (= file ``offset // file_size`` at
        ``offset % file_size``, continuing into the next files), without moving
        this writer's position; thread safe.  `sync_position` brings the
        writer's own position up to what was written this way."""
        data = memoryview(data).cast('B')
        done = 0
        while done < len(data):
            nr, inner = divmod(offset + done, self.file_size)
            n = min(len(data) - done, self.file_size - inner)
            fd = self._pwrite_fd(nr)
            at = inner
            part = data[done:done + n]
            while len(part):
                k = os.pwrite(fd, part, at)
                part, at = part[k:], at + k
            done += n
        return len(data)

    _pw_fds = None
    _pw_lock = None
    _pw_made = None         # files of the sequence that exist by this writer's doing

    def _pwrite_fd(self, nr):
        if self._pw_fds is None:
            import threading
            self._pw_fds, self._pw_lock = {}, threading.Lock()
        fd = self._pw_fds.get(nr)
        if fd is None:
            with self._pw_lock:
                fd = self._pw_fds.get(nr)
                if fd is None:
                    try:
                        name = _as_name(self.files[nr])
                    except IndexError:
                        raise OSError('ran out of files.') from None
                    # (the files this writer has opened itself exist -- 'w+b' made them -- and
                    # must not be truncated under it; later ones are created here, EMPTY the
                    # first time: what an older, longer file of that name held must not stay)
                    flags = os.O_WRONLY | os.O_CREAT
                    if self._pw_made is None:
                        self._pw_made = set(range(self.file_nr + 1))
                    if nr not in self._pw_made and nr > self.file_nr:
                        flags |= os.O_TRUNC
                    self._pw_made.add(nr)
                    fd = self._pw_fds[nr] = os.open(name, flags, 0o666)
        return fd

    def sync_position(self, total):
        """Everything up to stream byte `total` has been written positionally:
        close those descriptors and stand where a sequential writer would
        stand after the same bytes."""
        if self._pw_fds:
            for fd in self._pw_fds.values():
                os.close(fd)
            self._pw_fds = {}
        here = self._starts[self.file_nr] + self.fh.tell()
        if total <= here:
            return
        nr, inner = divmod(total, self.file_size)
        if inner == 0 and nr > 0:
            nr, inner = nr - 1, self.file_size          # (a full last file stays the current one, as after write())
        while len(self._starts) <= nr:
            self._starts.append(self._starts[-1] + self.file_size)
        if nr != self.file_nr:
            name = self.files[nr]
            fh = io.open(name, 'r+b')                   # (exists: written positionally; 'w+b' would truncate it)
            self.fh.close()
            self.fh, self.file_nr = fh, nr
        self.fh.seek(inner)

    def memmap(self, dtype=np.uint8, mode=None, offset=None, shape=None, order='C'):
        """Writable map of the next bytes of the current file (moving to the
        next file when the current one is full)."""
        if shape is None:
            raise ValueError('cannot make writable memmap without shape.')
        if self.closed:
            raise ValueError('memmap of closed file.')
        if offset is not None and offset != self.tell():
            raise OSError('a sequential writer cannot seek.')
        dtype = np.dtype(dtype)
        shape = shape if isinstance(shape, tuple) else (shape,)
        count = dtype.itemsize * int(np.prod(shape, dtype=np.int64))
        if self.file_size is not None:
            if self.fh.tell() == self.file_size:
                self._advance()
            if self.fh.tell() + count > self.file_size:
                raise ValueError('mmap length exceeds individual file size')
        inner = self.fh.tell()
        mm = np.memmap(self.fh, dtype, (mode or self.mode).replace('b', ''), inner, shape, order)
        self.fh.seek(inner + count)
        return mm


def open(files, mode='rb', file_size=None, opener=None):
    """Reader (``'r'`` in mode) or writer (``'w'``) over a file sequence
    (helpers/sequentialfile.py:381-425)."""
    if 'r' in mode:
        if file_size is not None:
            raise TypeError("cannot pass in 'file_size' for reading.")
        return SequentialFileReader(files, mode, opener=opener)
    if 'w' in mode:
        return SequentialFileWriter(files, mode, file_size=file_size, opener=opener)
    raise ValueError("invalid mode '{0}'".format(mode))
