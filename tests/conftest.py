import json
import os
import sys

import numpy as np
import pytest

# the readers create a placement arena on their first large output (a quarter
# of the free device memory by default): keep it small in the test process,
# which also runs full-size cases that want most of the GPU for themselves
os.environ.setdefault('BB_ARENA_GIB', '6')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


_launcher = None


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: test needs a real MI355X (run with -m gpu)")
    # tests/test_multiprocess_gpu.py needs rank processes whose PARENT never
    # initialised the GPU (a process that has must not fork + exec): start the
    # small helper that will start them NOW, before anything here touches the
    # GPU (`_has_gpu` below does).  It idles on stdin and ends with the session.
    global _launcher
    if _launcher is None and os.path.exists('/dev/kfd'):
        import subprocess
        try:
            _launcher = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'mp_launcher.py')],
                                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        except Exception:
            _launcher = None


def pytest_unconfigure(config):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.close()
            _launcher.wait(timeout=10)
        except Exception:
            try:
                _launcher.kill()
            except Exception:
                pass
        _launcher = None


def rank_launcher():
    """The helper process that starts rank processes (None without a GPU)."""
    return _launcher if _launcher is not None and _launcher.poll() is None else None


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="needs a GPU (torch.cuda.is_available() is False)")
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def manifest():
    with open(os.path.join(GOLD, 'manifest.json')) as f:
        return json.load(f)['cases']


@pytest.fixture(scope='session')
def levels_json():
    with open(os.path.join(GOLD, 'levels.json')) as f:
        return {k: np.array(v, dtype=np.uint32) for k, v in json.load(f).items()}


def golden_path(rel):
    return os.path.join(GOLD, rel)


def load_expected(name):
    return np.load(os.path.join(GOLD, 'expected', name + '.npz'))['data']


def load_file(rel):
    return np.fromfile(os.path.join(GOLD, rel), dtype=np.uint8)


def bits_equal(a, b):
    """Bit-exact comparison of float32/complex64 arrays."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return (a.shape == b.shape and a.dtype == b.dtype
            and np.array_equal(a.view(np.uint32), b.view(np.uint32)))


# -- product / experiment build (baseband_amd/_lib.py EXPERIMENTS) -------------
# The product library has ONE dispatch; the measurement variants of rounds 1-2
# (BB_TUNE_FLAT_VARIANT etc., include/bbdecode_exp.h) exist in the experiment
# build only (make -C baseband_amd/csrc EXPERIMENTS=1; BB_EXPERIMENTS=1 pytest ...).
def _experiments():
    from baseband_amd import _lib
    return _lib.EXPERIMENTS


def variants(*wanted):
    """Flat-kernel variants a test may force: all of `wanted` on the
    experiment build, the product dispatch (5) otherwise."""
    return tuple(wanted) if _experiments() else (5,)


def nt_modes():
    return (0, 1) if _experiments() else (1,)


def tune_exp(knob, value):
    """Set an experiment-only knob; a no-op on the product library (whose
    only state is the default the tests restore)."""
    if _experiments():
        from baseband_amd import kernels
        kernels.tune(knob, value)


needs_experiments = pytest.mark.skipif(
    os.environ.get('BB_EXPERIMENTS', '') in ('', '0'),
    reason="measurement variant: experiment build only (BB_EXPERIMENTS=1)")
