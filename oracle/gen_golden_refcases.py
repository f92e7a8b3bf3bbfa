"""Record the reference's outcomes for the behaviour cases (development
container only; test infrastructure).

    /opt/conda/bin/python3.9 -W ignore oracle/gen_golden_refcases.py [group ...]

Reads the operation lists of oracle/refcases/<group>.py, runs each case with
tests/casekit.Runner inside the REAL reference (mhvk/baseband imported from
/root/reference, astropy types), and writes tests/golden/refcases/<group>.json:
the operations and, next to them, what the reference answered for every one of
them.  Only those answers travel; tests/test_refcases*.py replays the
operations on this package and compares.
"""
import importlib
import json
import os
import shutil
import sys
import tempfile

if os.environ.get('PYTHONHASHSEED') != '0':        # (set orders show in a few messages: the same file every run)
    os.environ['PYTHONHASHSEED'] = '0'
    os.execv(sys.executable, [sys.executable, '-W', 'ignore'] + sys.argv)

import numpy as np

np.asscalar = getattr(np, 'asscalar', lambda a: a.item())
np.alen = getattr(np, 'alen', len)
sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import astropy.units as u                       # noqa: E402
from astropy.time import Time, TimeDelta        # noqa: E402
import baseband                                 # noqa: E402,F401
import baseband.data                            # noqa: E402

import casekit                                  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'refcases')
_EPOCH = Time('1970-01-01T00:00:00', scale='utc')


class RefUniverse(casekit.Universe):
    name = 'baseband (reference)'
    sample_dir = os.path.dirname(baseband.data.__file__)

    def module(self, name):
        if name in ('sequentialfile', 'sf'):
            return importlib.import_module('baseband.helpers.sequentialfile')
        if name == 'io':
            return importlib.import_module('baseband.io')
        if name == 'top':
            return importlib.import_module('baseband')
        return importlib.import_module('baseband.' + name)

    def sample(self, name):
        p = os.path.join(self.sample_dir, name)
        if not os.path.exists(p):
            raise FileNotFoundError(p)
        return p

    def time(self, iso):
        return Time(iso, scale='utc', precision=9)

    def duration_ns(self, ns):
        return ns * u.ns

    def rate_hz(self, hz):
        return hz * u.Hz

    def nbytes(self, n):
        return n * u.byte

    def unit(self, name):
        return 'time' if name == 'time' else u.Unit(name)

    def special(self, x):
        if isinstance(x, Time):
            if x.isscalar:
                return {'t': Time(x, precision=9).utc.isot}
            return [{'t': s} for s in Time(x, precision=9).utc.isot.tolist()]
        if isinstance(x, TimeDelta):
            return float(x.to_value(u.s))
        if isinstance(x, u.Quantity):
            if x.unit.is_equivalent(u.Hz):
                v = x.to_value(u.Hz)
            elif x.unit.is_equivalent(u.s):
                v = x.to_value(u.s)
            elif x.unit.is_equivalent(u.byte):
                v = x.to_value(u.byte)
            elif x.unit == u.dimensionless_unscaled:
                v = x.value
            else:
                return {'q': self._plain(x.value), 'unit': str(x.unit)}
            return self._plain(v)
        if isinstance(x, u.UnitBase):
            return {'unit': str(x)}
        return NotImplemented

    @staticmethod
    def _plain(v):
        v = np.asarray(v)
        if v.ndim == 0:
            f = float(v)
            return int(f) if f.is_integer() and abs(f) < 2 ** 53 and v.dtype.kind in 'iu' else f
        return [float(e) for e in v.reshape(-1)]


def record(group):
    mod = importlib.import_module('refcases.' + group)
    universe = RefUniverse()
    cases = []
    for c in mod.CASES:
        tmp = tempfile.mkdtemp(prefix='refcase_')
        runner = casekit.Runner(universe, tmp)
        try:
            outcomes = runner.run(c['steps'])
        finally:
            runner.finish()
            shutil.rmtree(tmp, ignore_errors=True)
        for o in outcomes:
            if o.get('raises') == 'Missing' or (o.get('raises') in ('AttributeError', 'TypeError', 'NameError')
                                                 and not c.get('expects_attribute_errors')):
                print('  !! {}: {} {}'.format(c['name'], o['raises'], o.get('msg', '')[:150]))
        entry = dict(c)
        entry['expect'] = outcomes
        cases.append(entry)
        print('  {}: {} outcomes, {} raise'.format(c['name'], len(outcomes), sum('raises' in o for o in outcomes)))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, group + '.json'), 'w') as f:
        json.dump({'made_by': 'oracle/gen_golden_refcases.py ' + group,
                   'reference': 'mhvk/baseband (/root/reference), astropy ' + __import__('astropy').__version__
                                + ', numpy ' + np.__version__,
                   'cases': cases}, f, indent=None, separators=(',', ':'))
        f.write('\n')


if __name__ == '__main__':
    groups = sys.argv[1:] or sorted(n[:-3] for n in os.listdir(os.path.join(HERE, 'refcases'))
                                    if n.endswith('.py') and not n.startswith('_'))
    for g in groups:
        print(g)
        record(g)
