"""Short constructors for the operations tests/casekit.py interprets."""


def S(name):
    """Path of a sample recording."""
    return {'$sample': name}


def SS(*names):
    return {'$samples': list(names)}


def T(name):
    """Path inside the case's scratch directory."""
    return {'$tmp': name}


def V(path):
    """Value of an earlier result (dotted path)."""
    return {'$': path}


def TIME(iso):
    return {'$time': iso}


def NS(ns):
    return {'$ns': ns}


def HZ(hz):
    return {'$hz': hz}


def NBYTES(n):
    return {'$nbytes': n}


def UNIT(name):
    return {'$unit': name}


def SL(*a):
    return {'$slice': list(a)}


def TUP(*a):
    return {'$tuple': list(a)}


def HEX(h):
    return {'$hex': h}


def FILL(byte, count):
    """`count` bytes of value `byte`."""
    return {'$fill': [byte, count]}


ELLIPSIS = {'$ellipsis': 1}


def ZEROS(shape, dt='f4'):
    return {'$zeros': list(shape), 'dt': dt}


def ARRAY(values, dt=None):
    if dt == 'c8':                      # complex values travel as [re, im] pairs on a last axis
        def pairs(v):
            return [pairs(e) for e in v] if isinstance(v, (list, tuple)) else [complex(v).real, complex(v).imag]
        values = pairs(values)
    return {'$array': values, 'dt': dt}


def RNG(seed, shape, levels=None, complex=False):
    d = {'$rng': seed, 'shape': list(shape)}
    if levels is not None:
        d['levels'] = levels
        d['complex'] = complex
    return d


def _opts(d, o):
    d.update({k: v for k, v in o.items() if v is not None})
    return d


def open_(as_, fmt, *args, quiet=None, msg=None, msg_has=None, any_warns=None, we_may_manage=None, **kw):
    return _opts({'op': 'open', 'as': as_, 'fmt': fmt, 'args': list(args), 'kw': kw},
                 dict(quiet=quiet, msg=msg, msg_has=msg_has, any_warns=any_warns, we_may_manage=we_may_manage))


def file_(as_, path, mode='rb'):
    return {'op': 'file', 'as': as_, 'path': path, 'mode': mode, 'quiet': True}


def call(as_, fn, *args, quiet=None, msg=None, any_warns=None, exact_exc=None, some_warns=None,
         we_may_manage=None, **kw):
    return _opts({'op': 'call', 'as': as_, 'fn': fn, 'args': list(args), 'kw': kw},
                 dict(quiet=quiet, msg=msg, any_warns=any_warns, exact_exc=exact_exc, some_warns=some_warns,
                      we_may_manage=we_may_manage))


def do(fn, *args, **kw):
    """A call whose return value is not of interest (exceptions and warnings still are)."""
    return call(None, fn, *args, quiet=True, **kw)


def get(of, as_=None, quiet=None, prefix=None):
    return _opts({'op': 'get', 'of': of, 'as': as_}, dict(quiet=quiet, prefix=prefix))


def gets(obj, *attrs):
    """One `get` per attribute of `obj`."""
    return [get(obj + '.' + a) for a in attrs]


def set_(of, to):
    return {'op': 'set', 'of': of, 'to': to}


def item(as_, of, key, quiet=None, prefix=None):
    return _opts({'op': 'item', 'as': as_, 'of': of, 'key': key}, dict(quiet=quiet, prefix=prefix))


def setitem(of, key, to):
    return {'op': 'setitem', 'of': of, 'key': key, 'to': to}


def let(as_, to):
    return {'op': 'let', 'as': as_, 'to': to, 'quiet': True}


def eq(a, b):
    return {'op': 'eq', 'a': a, 'b': b}


def fn(as_, name, *args, quiet=None):
    return _opts({'op': 'fn', 'as': as_, 'name': name, 'args': list(args)}, dict(quiet=quiet))


def digest(path):
    return {'op': 'digest', 'path': path}


def exists(path):
    return {'op': 'exists', 'path': path}


def listdir(as_=None):
    return {'op': 'listdir', 'as': as_}


def repeat(n, *steps):
    return {'op': 'repeat', 'n': n, 'steps': flat(steps)}


def each(var, in_, *steps):
    return {'op': 'each', 'var': var, 'in': in_, 'steps': flat(steps)}


def close(name):
    return do(name + '.close')


def flat(steps):
    out = []
    for s in steps:
        if isinstance(s, (list, tuple)):
            out.extend(flat(s))
        else:
            out.append(s)
    return out


def case(name, about, *steps, gpu=True):
    """`about`: the behaviour probed and where the reference tests it."""
    return {'name': name, 'about': about, 'gpu': gpu, 'steps': flat(steps)}
