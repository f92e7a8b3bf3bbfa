"""One ``open()`` for every format: resolves what `name` is (file handle, file
name, list of names, or a ``{file_nr}`` template), opens it -- a sequence goes
through `helpers.sequentialfile` -- and hands the handle to the format's file
or stream class.  Same call shapes as the reference's ``FileOpener``
(base/base.py:1650-1837): modes ``'rb'``, ``'rs'``, ``'ws'`` (``'r'``/``'w'``
mean the stream modes), ``file_size=`` for written sequences, template fields
filled from ``header0`` or the header keywords.
"""
import io
import os

from ..helpers import sequentialfile as sf
from .writer import LazyWriteFile
from .quantities import normalize_kwargs

__all__ = ['FormatOpener', 'source_kind']


def _is_device_bytes(name):
    """A tensor on the GPU: the file image itself, already staged in HBM."""
    import torch
    return isinstance(name, torch.Tensor)


def source_kind(name):
    """'fh', 'name', 'sequence' or 'template' (base/base.py:1694-1741);
    'device' for a GPU tensor holding the file bytes (resident.py)."""
    if _is_device_bytes(name):
        return 'device'
    if isinstance(name, str):
        return 'template' if ('{' in name and '}' in name) else 'name'
    if isinstance(name, os.PathLike):
        return 'name'
    if hasattr(name, 'read') or hasattr(name, 'write'):
        return 'fh'
    if hasattr(name, '__getitem__') or hasattr(name, '__iter__'):
        return 'sequence'
    raise ValueError("name '{}' not understood.".format(name))


def _reopen_stream(opener, source, kwargs, offset, closed=False):
    """Unpickling: open again from the recorded source and go to `offset`; a reader
    that was closed when pickled comes back closed."""
    reader = opener(source, 'rs', **kwargs)
    reader.offset = offset
    if closed:
        reader.close()
    return reader


class FormatOpener:
    """``open`` function of one format.

    Parameters
    ----------
    fmt : str
        Format name, for messages.
    classes : dict
        ``{'rb': FileReader, 'rs': StreamReader, 'ws': StreamWriter}``
        (whichever exist).
    sequencer : class
        `FileNameSequencer` flavour used for templates.
    default_file_size : callable, optional
        ``f(header0) -> int`` -- file size used for written sequences when
        the caller gives none (DADA, GUPPI: one frame per file).
    """
    def __init__(self, fmt, classes, sequencer=sf.FileNameSequencer,
                 default_file_size=None, adopt_header=None, header_keywords=None):
        self.fmt, self.classes = fmt, dict(classes)
        # ``f() -> set of lower-case names`` a header of this format can be built from:
        # keywords that only fill a file-name template are not passed on to a writer
        # (the reference pops what `header_class.fromvalues` can have used, the template
        # pops its own: base/base.py:1737-1779); None: any keyword may be the header's
        self.header_keywords = header_keywords
        self.sequencer = sequencer
        self.default_file_size = default_file_size
        # ``f(header) -> this package's header``: a ``header0=`` that is the
        # REFERENCE's header object (what its callers have in hand when they
        # write through the plugin seam) is rebuilt from its words / cards
        self.adopt_header = adopt_header

    def __reduce__(self):
        # the opener of a format is a module-level singleton named `open`
        return (_module_opener, (self.classes[next(iter(self.classes))].__module__,))

    def normalize_mode(self, mode):
        for candidate in (mode, mode[::-1], mode + 's' if mode in ('r', 'w') else None):
            if candidate in self.classes:
                return candidate
        raise ValueError("invalid mode: {} ({} supports {}).".format(
            mode, self.fmt, sorted(self.classes)))

    # -- templates
    def _sequencer_for(self, template, mode, kwargs):
        """Template -> file-name sequencer.  Fields are filled from ``header0``
        (if given) and the keywords; when reading, keywords the template used
        are removed so that they do not reach the reader class."""
        values = {}
        header0 = kwargs.get('header0')
        if header0 is not None:
            values.update({k: header0[k] for k in header0.keys()})
        values.update(kwargs)
        values = _CaseBlind(values)
        fns = self.sequencer(template, values)
        if mode[0] == 'r':
            for key in values.consulted.intersection(kwargs):
                kwargs.pop(key)
        elif self.header_keywords is not None:
            known = self.header_keywords()
            for key in values.consulted.intersection(kwargs):
                if key.lower() not in known:
                    kwargs.pop(key)
        return fns

    # -- handles
    def _handle(self, name, mode, kwargs):
        kind = source_kind(name)
        if kind == 'fh':
            # a handle the caller opened: a reader on it can still be pickled when the
            # handle says where its bytes come from -- a sequence of named files
            # (helpers.sequentialfile) or one named file -- as the reference's readers
            # can (base/base.py:123-151: the file name and position travel)
            source = None
            if mode[0] == 'r':
                files = getattr(name, 'files', None)
                if isinstance(name, sf.SequentialFileReader) and files is not None \
                        and all(isinstance(f, (str, os.PathLike)) for f in
                                (files if isinstance(files, (list, tuple)) else [])):
                    source = list(files) if isinstance(files, (list, tuple)) else None
                elif isinstance(getattr(name, 'name', None), (str, os.PathLike)) and os.path.exists(name.name):
                    source = os.fspath(name.name)
            return name, source
        if kind == 'device':
            if mode[0] != 'r':
                raise ValueError("a device tensor can only be opened for reading.")
            from ..resident import DeviceFile
            return DeviceFile(name), None
        if kind == 'template':
            name, kind = self._sequencer_for(name, mode, kwargs), 'sequence'
        if mode[0] == 'r':
            if kind == 'sequence':
                return sf.open(name, 'rb'), name
            return io.open(name, 'rb'), os.fspath(name)
        if kind == 'sequence':
            if mode == 'wb':
                raise ValueError("{} does not support writing to a sequence or "
                                 "template in binary mode.".format(self.fmt))
            return sf.open(name, 'w+b', file_size=kwargs.pop('file_size', None)), name
        return LazyWriteFile(name), os.fspath(name)

    def __call__(self, name, mode='rs', **kwargs):
        mode = self.normalize_mode(mode)
        # what reference callers pass (Quantity rates and sizes, Time instants:
        # vdif/base.py:422-454 there) -> plain Hz / bytes / datetime64[ns]
        kwargs = normalize_kwargs(kwargs)
        if self.adopt_header is not None and kwargs.get('header0') is not None:
            kwargs['header0'] = self.adopt_header(kwargs['header0'])
        if (mode == 'ws' and self.default_file_size is not None
                and source_kind(name) in ('sequence', 'template')
                and 'file_size' not in kwargs and kwargs.get('header0') is not None):
            kwargs['file_size'] = self.default_file_size(kwargs['header0'])
        fh, source = self._handle(name, mode, kwargs)
        init_args = dict(kwargs)
        try:
            opened = self.classes[mode](fh, **kwargs)
        except Exception:
            if fh is not name:
                try:
                    getattr(fh, 'discard', fh.close)()
                except Exception:
                    pass
            raise
        if mode == 'rs' and source is not None:
            # what GPUStreamReaderBase.__reduce__ needs to come back after pickling
            opened._pickle_recipe = (_reopen_stream, self, source, init_args)
        if mode == 'rs':
            _prepare_output_memory(opened)
        return opened


def _prepare_output_memory(reader):
    """A stream whose decoded samples are 1 GiB and more: let the output arena
    start taking its memory now, in the background, instead of inside the first
    large ``read()`` (placement.prepare_output; a no-op without a GPU, with
    BB_ARENA=0 / BB_ARENA_PREPARE=0, or when the arena has room already).

    The size comes from what ``open()`` knows WITHOUT searching the file for
    its last header (that search, and its warnings, stay lazy: ADVICE r5): the
    sample count if the format has set it, else the bytes of the file times
    the expansion of one frame -- an upper bound, which is all a growth step
    needs.  A process that opens large streams only to read small pieces sets
    BB_ARENA_PREPARE=0 (or BB_ARENA=0): the step is taken when a reader of a
    >= 1 GiB stream opens, not when its first large read arrives."""
    try:
        item = 8 if reader.complex_data else 4
        for dim in reader.sample_shape:
            item *= int(dim)
        nsample = reader.__dict__.get('_nsample_found')
        if nsample is None:
            nsample = _nsample_bound(reader)
        if nsample is not None and nsample * item >= (1 << 30):
            from ..placement import prepare_output
            prepare_output(int(nsample) * item)
    except Exception:
        pass


def _nsample_bound(reader):
    """Samples the file of `reader` can hold at most, from its size alone."""
    raw = getattr(reader, 'fh_raw', None)
    try:
        set_nbytes = int(reader._set_nbytes)
        spf = int(reader.samples_per_frame)
        here = raw.tell()
        size = raw.seek(0, 2)
        raw.seek(here)
    except Exception:
        return None
    if set_nbytes <= 0:
        return None
    return (size // set_nbytes) * spf


class _CaseBlind(dict):
    """Header values looked up by a template field, ignoring case; remembers
    which of its keys were consulted."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.consulted = set()

    def __getitem__(self, key):
        for k in self:
            if k == key or (isinstance(k, str) and k.lower() == key.lower()):
                self.consulted.add(k)
                return super().__getitem__(k)
        raise KeyError(key)

    def __contains__(self, key):
        return any(k == key or (isinstance(k, str) and k.lower() == key.lower())
                   for k in self)


def _module_opener(module_name):
    import importlib
    return importlib.import_module(module_name).open
