"""Recorded behaviour cases: a small interpreter of operation lists.

A *case* is data: an ordered list of operations on readers, writers, frames,
headers and plain files ("open this sample as a stream, read 12 samples, seek
to a time, write what was read with that header, ...").  The interpreter runs
a case inside a *universe* -- the real reference (astropy types; only in the
development container, oracle/gen_golden_refcases.py) or this package (plain
Hz, numpy.datetime64, device tensors; `AmdUniverse` below) -- and reduces the
outcome of EVERY operation to plain JSON: values, shapes, digests of arrays and
of written files, header words, the class of an exception, the categories of
the warnings.  The generator stores the reference's outcomes next to the
operations in tests/golden/refcases/*.json; tests/test_refcases*.py runs the same
operations here and compares outcome by outcome (`compare`).

Nothing here is reference source: the operation lists are written for this
repository (oracle/refcases/*.py), the outcomes are the reference's ANSWERS.
Test infrastructure only -- nothing in baseband_amd imports this module.
"""
import hashlib
import io
import math
import os
import re
import warnings

import numpy as np

__all__ = ['Runner', 'AmdUniverse', 'compare', 'load_group']

_BUILTIN_SKIP = ('Exception', 'BaseException', 'object')


def sha256(b):
    return hashlib.sha256(b).hexdigest()


# ---------------------------------------------------------------- universes
class Universe:
    """What differs between the two packages: where the modules are, how an
    instant / a duration / a rate is made, and how their results look."""
    name = '?'

    def module(self, name):
        raise NotImplementedError

    def sample(self, name):
        raise NotImplementedError

    def time(self, iso):
        raise NotImplementedError

    def duration_ns(self, ns):
        raise NotImplementedError

    def rate_hz(self, hz):
        raise NotImplementedError

    def nbytes(self, n):
        return n

    def host(self, x):
        """Array-like of the universe -> numpy array (or x itself)."""
        return x

    def special(self, x):
        """JSON form of a value only this universe knows (Time, Quantity...);
        NotImplemented for everything else."""
        return NotImplemented

    def unit(self, name):
        """Argument for ``tell(unit=...)``."""
        return name

    def package_dirs(self):
        """Directories whose modules' warnings count as the package's own."""
        mod = self.module('vdif')
        return [os.path.dirname(os.path.dirname(os.path.abspath(mod.__file__))) + os.sep]


class _UnitOfTime:
    """What ``fh.tell(unit)`` needs of a caller's unit object (astropy's ``u.ms``): ``to('s')``
    and ``number * unit``.  The product is reduced to SECONDS at once -- the form durations
    take in recorded outcomes -- instead of a Quantity."""
    _S = {'s': 1.0, 'ms': 1e-3, 'us': 1e-6, 'ns': 1e-9, 'min': 60.0}

    def __init__(self, name):
        self.name, self.s = name, self._S[name]

    def to(self, other):
        return self.s / self._S[str(other)]

    def __rmul__(self, number):
        return float(number) * self.s


class AmdUniverse(Universe):
    """This package: rates in Hz, instants numpy.datetime64[ns], durations
    numpy.timedelta64 or seconds, decoded samples torch tensors."""
    name = 'baseband_amd'

    def __init__(self, sample_dir):
        self.sample_dir = sample_dir

    def module(self, name):
        import importlib
        if name in ('sequentialfile', 'sf'):
            return importlib.import_module('baseband_amd.helpers.sequentialfile')
        if name == 'io':
            return importlib.import_module('baseband_amd.io')
        if name == 'top':
            return importlib.import_module('baseband_amd')
        return importlib.import_module('baseband_amd.' + name)

    def sample(self, name):
        return os.path.join(self.sample_dir, name)

    def time(self, iso):
        if 'T23:59:60' in iso:                  # inside a leap second: numpy has no label for it
            from baseband_amd.base.quantities import LeapSecondInstant
            return LeapSecondInstant.fromisot(iso)
        return np.datetime64(iso, 'ns')

    def duration_ns(self, ns):
        return np.timedelta64(int(ns), 'ns')

    def rate_hz(self, hz):
        return float(hz)

    def unit(self, name):
        return name if name in ('time', 's') or name not in _UnitOfTime._S else _UnitOfTime(name)

    def host(self, x):
        try:
            import torch
        except ImportError:             # pragma: no cover
            return x
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
        return x

    def special(self, x):
        if isinstance(x, np.datetime64):
            return {'t': str(x.astype('datetime64[ns]'))}
        if type(x).__name__ == 'LeapSecondInstant':
            return {'t': str(x)}
        if isinstance(x, np.timedelta64):
            return float(x.astype('timedelta64[ns]').astype(np.int64)) * 1e-9
        try:
            import torch
        except ImportError:             # pragma: no cover
            return NotImplemented
        if isinstance(x, torch.dtype):
            return {'dtype': str(x).replace('torch.', '')}
        if isinstance(x, torch.Size):
            return [int(v) for v in x]
        return NotImplemented


# ---------------------------------------------------------------- the runner
_TOKEN = re.compile(r"\.?([A-Za-z_][A-Za-z_0-9]*)|\[(-?\d+)\]|\['([^']*)'\]")


class Missing(KeyError):
    """A name an earlier (failed) operation should have made."""


class Runner:
    MODULES = ('vdif', 'mark4', 'mark5b', 'dada', 'guppi', 'gsb', 'sequentialfile', 'sf', 'io', 'top', 'base')

    def __init__(self, universe, tmpdir):
        self.u = universe
        self.tmp = str(tmpdir)
        self.vars = {}
        self.opened = []

    # ---- values written in a case -> objects
    def val(self, x):
        if isinstance(x, list):
            return [self.val(v) for v in x]
        if not isinstance(x, dict):
            return x
        if len(x) == 1:
            (k, v), = x.items()
            if k == '$':
                return self.path(v)
            if k == '$sample':
                return self.u.sample(v)
            if k == '$samples':
                return [self.u.sample(n) for n in v]
            if k == '$tmp':
                return os.path.join(self.tmp, v)
            if k == '$time':
                return self.u.time(v)
            if k == '$ns':
                return self.u.duration_ns(v)
            if k == '$hz':
                return self.u.rate_hz(v)
            if k == '$nbytes':
                return self.u.nbytes(v)
            if k == '$unit':
                return self.u.unit(v)
            if k == '$slice':
                return slice(*[self.val(e) for e in v])
            if k == '$tuple':
                return tuple(self.val(e) for e in v)
            if k == '$set':
                return set(self.val(e) for e in v)
            if k == '$hex':
                return bytes.fromhex(v)
            if k == '$fill':
                return bytes([v[0]]) * v[1]
            if k == '$ellipsis':
                return Ellipsis
            if k == '$dtype':
                return np.dtype(v)
            if k == '$bytesio':
                return io.BytesIO(self.val(v) if v is not None else b'')
        if '$zeros' in x:
            return np.zeros(tuple(x['$zeros']), dtype=x.get('dt', 'f4'))
        if '$array' in x:
            if x.get('dt') == 'c8':         # complex values travel as [re, im] pairs on a last axis
                a = np.array(x['$array'], dtype='f4')
                return np.ascontiguousarray(a).view('c8')[..., 0]
            return np.array(x['$array'], dtype=x.get('dt'))
        if '$rng' in x:
            # own generator (the same numbers under any numpy): a 64-bit LCG stepped per element
            n = int(np.prod(x['shape']))
            with np.errstate(over='ignore'):
                state = (np.arange(1, n + 1, dtype=np.uint64) + np.uint64(x['$rng'])) * np.uint64(6364136223846793005) \
                    + np.uint64(1442695040888963407)
                state ^= state >> np.uint64(29)
                state *= np.uint64(0xBF58476D1CE4E5B9)
                bits = (state >> np.uint64(40)).astype(np.int64)
            if 'levels' in x:
                lev = np.array(x['levels'], dtype='f4')
                out = lev[bits % len(lev)]
                if x.get('complex'):
                    return (out + 1j * lev[(bits >> 8) % len(lev)]).astype('c8').reshape(x['shape'])
                return out.reshape(x['shape'])
            return (bits & 0xff).astype(np.uint8).reshape(x['shape'])
        return {k: self.val(v) for k, v in x.items()}

    def path(self, text):
        """'fr.header0.time', 'fs.frames[0].header', "h['frame_nr']", 'vdif.VDIFHeader'."""
        pos, obj, first = 0, None, True
        while pos < len(text):
            m = _TOKEN.match(text, pos)
            if m is None:
                raise ValueError('cannot read path {!r} at {}'.format(text, pos))
            name, index, key = m.groups()
            if first:
                if name is None:
                    raise ValueError('path must start with a name: ' + text)
                if name in self.vars:
                    obj = self.vars[name]
                elif name in self.MODULES:
                    obj = self.u.module(name)
                else:
                    raise Missing(name)
                first = False
            elif name is not None:
                obj = getattr(obj, name)
            elif index is not None:
                obj = obj[int(index)]
            else:
                obj = obj[key]
            pos = m.end()
        return obj

    def _parent(self, text):
        """(object, last attribute) of a dotted path, for assignments."""
        head, _, last = text.rpartition('.')
        return self.path(head), last

    # ---- results -> JSON
    def norm(self, x, depth=0):
        if x is None or isinstance(x, (bool, str)):
            return x
        if depth > 6:
            return {'object': type(x).__name__}
        sp = self.u.special(x)
        if sp is not NotImplemented:
            return sp
        x = self.u.host(x)
        if isinstance(x, (bool, np.bool_)):
            return bool(x)
        if isinstance(x, (int, np.integer)):
            return int(x)
        if isinstance(x, (float, np.floating)):
            v = float(x)
            return v if math.isfinite(v) else {'float': repr(v)}
        if isinstance(x, (complex, np.complexfloating)):
            return {'c': [float(x.real), float(x.imag)]}
        if isinstance(x, (bytes, bytearray, memoryview)):
            b = bytes(x)
            return {'hex': b.hex()} if len(b) <= 24 else {'bytes': len(b), 'sha': sha256(b)}
        if isinstance(x, np.dtype):
            return {'dtype': x.name if x.names is None else str(x)}
        if isinstance(x, type):
            return {'type': x.__name__}
        if isinstance(x, slice):
            return {'slice': [x.start, x.stop, x.step]}
        if isinstance(x, np.ndarray):
            return self._norm_array(x)
        if isinstance(x, (list, tuple)):
            return [self.norm(v, depth + 1) for v in x]
        if isinstance(x, (set, frozenset)):
            return {'set': sorted((self.norm(v, depth + 1) for v in x), key=repr)}
        if isinstance(x, dict) and not hasattr(x, 'payload_nbytes'):
            return {str(k): self.norm(v, depth + 1) for k, v in x.items()}
        return self._norm_object(x, depth)

    def _norm_array(self, a):
        if a.dtype == object:
            return [self.norm(v) for v in a.tolist()]
        if a.dtype.names is not None:
            return {'shape': list(a.shape), 'dtype': str(a.dtype), 'sha': sha256(np.ascontiguousarray(a).tobytes())}
        if a.ndim == 0:
            return self.norm(a[()])
        kind = a.dtype.name
        out = {'shape': list(a.shape), 'dtype': kind, 'sha': sha256(np.ascontiguousarray(a).tobytes())}
        flat = a.reshape(-1)[:6]
        if a.dtype.kind == 'c':
            out['head'] = [[float(v.real), float(v.imag)] for v in flat]
        elif a.dtype.kind in 'fiub':
            out['head'] = [self.norm(v) for v in flat]
        return out

    def _norm_object(self, x, depth):
        cls = type(x).__name__
        # frame sets, frames, payloads, headers: by what they carry
        frames = getattr(x, 'frames', None)
        if frames is not None and not callable(frames) and hasattr(x, 'header0'):
            return {'frameset': cls, 'frames': [self.norm(f, depth + 1) for f in frames]}
        if hasattr(x, 'header') and hasattr(x, 'payload') and not hasattr(x, 'read'):
            out = {'frame': cls, 'header': self.norm(x.header, depth + 1), 'payload': self.norm(x.payload, depth + 1)}
            if hasattr(x, 'valid'):
                try:
                    out['valid'] = bool(x.valid)
                except Exception:
                    pass
            return out
        words = getattr(x, 'words', None)
        if words is not None and hasattr(x, 'keys') and len(words) and isinstance(words[0], str):
            return {'header': cls, 'words': [str(w) for w in words]}   # (GSB: the fields of a time-stamp line)
        if words is not None and hasattr(x, 'keys'):                    # binary headers: their words
            w = np.asarray(self.u.host(words)).astype(np.uint64)
            return {'header': cls, 'words': [int(v) for v in w.reshape(-1)] if w.size <= 40
                    else {'shape': list(w.shape), 'sha': sha256(np.ascontiguousarray(w).tobytes())}}
        if words is not None and hasattr(x, 'sample_shape') and hasattr(x, 'bps'):
            w = np.ascontiguousarray(self.u.host(words))
            return {'payload': cls, 'nbytes': int(w.nbytes), 'words_sha': sha256(w.tobytes()), 'bps': int(x.bps),
                    'complex': bool(x.complex_data), 'sample_shape': [int(v) for v in x.sample_shape]}
        if hasattr(x, 'keys') and hasattr(x, 'payload_nbytes'):          # ASCII headers (DADA, GUPPI)
            cards = {}
            for k in x.keys():
                if k in ('COMMENT', 'HISTORY', '') or str(k).startswith('_'):   # (comment lines, keyed by position)
                    continue
                try:
                    cards[str(k)] = self._card(x[k])
                except Exception as exc:            # pragma: no cover
                    cards[str(k)] = {'raises': type(exc).__name__}
            return {'header': cls, 'cards': cards}
        if hasattr(x, '_fields') and isinstance(x, tuple):               # (never reached: tuples handled above)
            return [self.norm(v, depth + 1) for v in x]
        if cls.endswith('Info'):                # (info objects: their answers are asked for one by one)
            cls = 'Info'
        return {'object': cls}

    @staticmethod
    def _card(v):
        if isinstance(v, (bool, np.bool_)):
            return bool(v)
        if isinstance(v, (int, np.integer)):
            return int(v)
        if isinstance(v, (float, np.floating)):
            return float(v)
        return str(v).strip()

    def _exc(self, exc):
        out = {'raises': type(exc).__name__}
        for c in type(exc).__mro__:
            if c.__module__ == 'builtins' and c.__name__ not in _BUILTIN_SKIP:
                out['builtin'] = c.__name__
                break
        out['msg'] = str(exc)[:300]
        return out

    # ---- operations
    def run(self, steps):
        out = []
        for st in steps:
            out.extend(self._step(st))
        return out

    def finish(self):
        for h in reversed(self.opened):
            try:
                h.close()
            except Exception:
                pass
        self.opened = []
        self.vars = {}

    def _step(self, st):
        op = st['op']
        if op == 'repeat':
            res = []
            for _ in range(st['n']):
                res.extend(self.run(st['steps']))
            return res
        if op == 'each':
            res = []
            for item in list(self.val(st['in'])):
                self.vars[st['var']] = item
                res.extend(self.run(st['steps']))
            return res
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            try:
                value = getattr(self, '_op_' + op)(st)
                if st.get('as'):
                    self.vars[st['as']] = value
                try:
                    rec = {'v': None if st.get('quiet') else self.norm(value)}
                except Exception as exc:                # (a fault of this module, not of the package)
                    rec = {'raises': 'CasekitNormError', 'msg': '{}: {}'.format(type(exc).__name__, exc)[:200]}
            except Exception as exc:                    # the outcome IS the exception
                rec = self._exc(exc)
                if st.get('as') and st['as'] not in self.vars:
                    self.vars.pop(st['as'], None)
        # (not the libraries' own: astropy / erfa grumbling about far-away years is not behaviour to match)
        seen = [w for w in caught if not issubclass(w.category, (DeprecationWarning, PendingDeprecationWarning,
                                                                   ResourceWarning, ImportWarning, FutureWarning))
                and w.category.__module__.split('.')[0] not in ('erfa', 'astropy', 'numpy', 'torch')
                and self._ours(w.filename)]
        if seen:
            rec['warns'] = [[w.category.__name__, str(w.message)[:200]] for w in seen]
        return [rec]

    def _ours(self, filename):
        """A warning raised by the package under test (or through it, at this interpreter's
        call): imports of third parties that happen during a case are not its behaviour."""
        f = os.path.abspath(filename)
        return f == os.path.abspath(__file__) or any(f.startswith(d) for d in self.u.package_dirs())

    def _op_open(self, st):
        mod = self.u.module(st['fmt'])
        fh = mod.open(*self.val(st.get('args', [])), **self.val(st.get('kw', {})))
        self.opened.append(fh)
        return fh

    def _op_file(self, st):
        fh = open(self.val(st['path']), st.get('mode', 'rb'))
        self.opened.append(fh)
        return fh

    def _op_call(self, st):
        fn = self.path(st['fn'])
        return fn(*self.val(st.get('args', [])), **self.val(st.get('kw', {})))

    def _op_get(self, st):
        return self.path(st['of'])

    def _op_set(self, st):
        obj, attr = self._parent(st['of'])
        setattr(obj, attr, self.val(st['to']))

    def _op_item(self, st):
        return self.path(st['of'])[self.val(st['key'])]

    def _op_setitem(self, st):
        self.path(st['of'])[self.val(st['key'])] = self.val(st['to'])

    def _op_let(self, st):
        return self.val(st['to'])

    def _op_del(self, st):
        self.vars.pop(st['name'], None)

    def _op_eq(self, st):
        a, b = self.val(st['a']), self.val(st['b'])
        a, b = self.u.host(a), self.u.host(b)
        r = a == b
        if isinstance(r, np.ndarray) or hasattr(r, 'all'):
            r = bool(np.asarray(self.u.host(r)).all()) and np.shape(a) == np.shape(b)
        return bool(r)

    def _op_digest(self, st):
        with open(self.val(st['path']), 'rb') as f:
            b = f.read()
        return {'size': len(b), 'sha': sha256(b)}

    def _op_exists(self, st):
        return os.path.exists(self.val(st['path']))

    def _op_listdir(self, st):
        return sorted(os.listdir(self.tmp))

    def _op_fn(self, st):
        args = self.val(st.get('args', []))
        return getattr(self, '_fn_' + st['name'])(*args)

    # ---- helper functions a case may name (op 'fn')
    def _fn_host(self, x):
        return np.asarray(self.u.host(x))

    def _fn_concat(self, parts, axis=0):
        return np.concatenate([np.asarray(self.u.host(p)) for p in parts], axis=axis)

    def _fn_truth(self, x):
        return bool(x)

    def _fn_len(self, x):
        return len(x)

    def _fn_add(self, a, b):
        return a + b

    def _fn_sub(self, a, b):
        return a - b

    def _fn_mul(self, a, b):
        return a * b

    def _fn_neg(self, a):
        return -a

    def _fn_abs_lt(self, a, b, tol_ns):
        d = self.norm(a - b)
        return abs(d) < tol_ns * 1e-9

    def _fn_as_int(self, x):
        return np.asarray(self.u.host(x)).astype(int)

    def _fn_allclose_to(self, x, value):
        return bool((np.asarray(self.u.host(x)) == value).all())

    def _fn_isinstance(self, x, clspath):
        return isinstance(x, self.path(clspath))

    def _fn_pickle_roundtrip(self, x):
        import pickle
        y = pickle.loads(pickle.dumps(x))
        if hasattr(y, 'close'):
            self.opened.append(y)
        return y

    def _fn_copy(self, x):
        import copy
        return copy.copy(x)

    def _fn_join(self, parts):
        return b''.join(parts)

    def _fn_list(self, x):
        return list(x)

    def _fn_sorted_keys(self, x):
        return sorted(str(k) for k in x.keys())

    def _fn_patch_file(self, path, offset, data):
        with open(path, 'r+b') as f:
            f.seek(offset)
            f.write(data)
        return os.path.getsize(path)

    def _fn_write_file(self, path, parts):
        with open(path, 'wb') as f:
            for p in parts:
                f.write(p)
        return os.path.getsize(path)

    def _fn_file_bytes(self, path, start=0, stop=None):
        with open(path, 'rb') as f:
            f.seek(start)
            return f.read() if stop is None else f.read(stop - start)

    def _fn_truncate(self, path, size):
        with open(path, 'r+b') as f:
            f.truncate(size)
        return os.path.getsize(path)


# ---------------------------------------------------------------- comparing
def _parse_t(s):
    return int(np.datetime64(s, 'ns').astype(np.int64))


def _same(want, got, where, out):
    if isinstance(want, dict) and isinstance(got, dict):
        if set(want) == {'t'} and set(got) == {'t'}:
            try:
                if abs(_parse_t(want['t']) - _parse_t(got['t'])) > 1:       # instants: to the nanosecond
                    out.append('{}: time {} != {}'.format(where, got['t'], want['t']))
            except Exception:
                if want['t'] != got['t']:
                    out.append('{}: time {} != {}'.format(where, got['t'], want['t']))
            return
        for k in want:
            if k == 'head':
                continue
            if k not in got:
                out.append('{}: no {!r} (have {})'.format(where, k, sorted(got)))
            else:
                _same(want[k], got[k], where + '.' + k, out)
        for k in got:
            if k not in want and k != 'head':
                out.append('{}: extra {!r}'.format(where, k))
        return
    if isinstance(want, list) and isinstance(got, list):
        if len(want) != len(got):
            out.append('{}: {} items, expected {}'.format(where, len(got), len(want)))
            return
        for i, (w, g) in enumerate(zip(want, got)):
            _same(w, g, '{}[{}]'.format(where, i), out)
        return
    if isinstance(want, bool) or isinstance(got, bool):
        if want is not got:
            out.append('{}: {!r} != {!r}'.format(where, got, want))
        return
    if isinstance(want, (int, float)) and isinstance(got, (int, float)):
        if isinstance(want, int) and isinstance(got, int):
            ok = want == got
        else:
            ok = math.isclose(want, got, rel_tol=1e-8, abs_tol=1e-12)
        if not ok:
            out.append('{}: {!r} != {!r}'.format(where, got, want))
        return
    if want != got:
        out.append('{}: {!r} != {!r}'.format(where, got, want))


def compare(steps, expected, got):
    """Differences between the reference's outcomes and this package's, one
    line each ([] = the case passes).  Exceptions: the same class, or both
    classes private to their package with the same builtin ancestor;
    messages only where the operation asks (``"msg": true``).  Warnings: the
    same categories in the same order (``"some_warns"``: only whether there are
    any; ``"any_warns"``: not compared).  ``"we_may_manage"``: the reference's
    exception is a limit of its search, and succeeding instead is accepted.  Instants to 1 ns; floats to 1e-8 (the
    reference derives durations from two-double Julian dates, this package from
    integer nanoseconds); integers, digests and words exactly."""
    flat = []

    def walk(sts):
        for st in sts:
            if st['op'] == 'repeat':
                for _ in range(st['n']):
                    walk(st['steps'])
            elif st['op'] == 'each':
                flat.append(None)           # (length depends on the data: matched by count below)
            else:
                flat.append(st)
    walk(steps)
    diffs = []
    if len(expected) != len(got):
        diffs.append('{} outcomes, expected {}'.format(len(got), len(expected)))
    labelled = flat if None not in flat and len(flat) == len(expected) else [None] * len(expected)
    for i, (w, g) in enumerate(zip(expected, got)):
        st = labelled[i] if i < len(labelled) else None
        where = '#{} {}'.format(i, _label(st))
        if 'raises' in w or 'raises' in g:
            if 'raises' not in g and st is not None and st.get('we_may_manage'):
                pass        # (a documented leniency: the reference gives up, this package need not)
            elif 'raises' not in g:
                diffs.append('{}: no exception, expected {} ({})'.format(where, w['raises'], w.get('msg', '')[:80]))
            elif 'raises' not in w:
                diffs.append('{}: raised {}: {}'.format(where, g['raises'], g.get('msg', '')[:160]))
            else:
                same_cls = w['raises'] == g['raises'] or (
                    w.get('builtin') == g.get('builtin') and w.get('builtin') is not None
                    and (st is None or not st.get('exact_exc')))
                if not same_cls:
                    diffs.append('{}: raised {} ({}), expected {} ({})'.format(
                        where, g['raises'], g.get('builtin'), w['raises'], w.get('builtin')))
                elif st is not None and st.get('msg') and w.get('msg') != g.get('msg'):
                    diffs.append('{}: message {!r}, expected {!r}'.format(where, g.get('msg'), w.get('msg')))
                elif st is not None and st.get('msg_has') and not (
                        st['msg_has'] in w.get('msg', '') and st['msg_has'] in g.get('msg', '')):
                    diffs.append('{}: message {!r} lacks {!r} (reference: {!r})'.format(
                        where, g.get('msg'), st['msg_has'], w.get('msg')))
        elif st is not None and st.get('prefix'):
            n = st['prefix']                            # (texts that go on to quote a library's own message)

            def cut(x):
                if isinstance(x, str):
                    return x[:n]
                if isinstance(x, dict):
                    return {k: cut(v) for k, v in x.items()}
                if isinstance(x, list):
                    return [cut(v) for v in x]
                return x
            _same(cut(w.get('v')), cut(g.get('v')), where, diffs)
        elif not (st is not None and st.get('quiet')):
            _same(w.get('v'), g.get('v'), where, diffs)
        if 'raises' in w and 'raises' not in g and st is not None and st.get('we_may_manage'):
            pass            # (nothing to compare the warnings of a read that went through with)
        elif st is not None and st.get('some_warns'):     # (that there are warnings, not how many)
            if bool(w.get('warns')) != bool(g.get('warns')):
                diffs.append('{}: warnings {}, expected {}'.format(
                    where, [m[:80] for _, m in g.get('warns', [])], [m[:80] for _, m in w.get('warns', [])]))
        elif st is None or not st.get('any_warns'):
            ww = [c for c, _ in w.get('warns', [])]
            gw = [c for c, _ in g.get('warns', [])]
            if ww != gw:
                diffs.append('{}: warnings {}, expected {} {}'.format(
                    where, gw, ww, [m[:80] for _, m in w.get('warns', [])]))
    return diffs


def _label(st):
    if st is None:
        return ''
    bits = [st['op']]
    for k in ('fn', 'of', 'fmt', 'name'):
        if k in st:
            bits.append(str(st[k]))
    if st.get('as'):
        bits.append('-> ' + st['as'])
    return ' '.join(bits)


def load_group(path):
    import json
    with open(path) as f:
        return json.load(f)
