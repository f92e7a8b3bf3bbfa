"""Behaviours of the reference, recorded as data, replayed on this package.

tests/golden/refcases/<group>.json holds, per case, a list of operations (made
for this repository: oracle/refcases/<group>.py) and the outcome the REAL
reference gave for each one of them (oracle/gen_golden_refcases.py, run in the
development container).  Every case is replayed here through
`casekit.Runner` on baseband_amd -- readers and writers decode and encode on
the GPU through libbbdecode.so -- and compared outcome by outcome: values,
shapes and SHA-256 digests of decoded arrays and written files, header words,
exception classes, warning categories.
"""
import glob
import json
import os

import pytest

import casekit
from conftest import GOLD as GOLDEN

GROUPS = sorted(glob.glob(os.path.join(GOLDEN, 'refcases', '*.json')))


def _cases():
    for path in GROUPS:
        group = os.path.basename(path)[:-5]
        for c in casekit.load_group(path)['cases']:
            yield pytest.param(c, id=group + ':' + c['name'],
                               marks=[pytest.mark.gpu] if c.get('gpu', True) else [])


@pytest.mark.parametrize('case', list(_cases()))
def test_case_replays_as_recorded(case, tmp_path):
    runner = casekit.Runner(casekit.AmdUniverse(os.path.join(GOLDEN, 'samples')), tmp_path)
    try:
        got = runner.run(case['steps'])
    finally:
        runner.finish()
    diffs = casekit.compare(case['steps'], case['expect'], got)
    report = os.environ.get('BB_REFCASE_REPORT')
    if report and diffs:
        with open(report, 'a') as f:
            f.write(json.dumps({'case': case['name'], 'about': case['about'][:80], 'diffs': diffs[:40]}) + '\n')
    assert not diffs, '\n'.join([case['about']] + diffs[:25])
