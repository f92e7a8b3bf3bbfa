"""`RawOffsets` (baseband_amd/base/offsets.py) against what the reference's class
answered for the same assignments (tests/golden/raw_offsets_cases.json, made by
oracle/gen_golden_offsets.py), and the cases of the reference's own unit test
(base/tests/test_offsets.py)."""
import json

import numpy as np
import pytest

from conftest import golden_path
from baseband_amd.base.offsets import RawOffsets

with open(golden_path('raw_offsets_cases.json')) as _f:
    GOLD = json.load(_f)


@pytest.mark.parametrize('trial', range(len(GOLD['fuzz'])))
def test_assignments_as_the_reference(trial):
    case = GOLD['fuzz'][trial]
    table = RawOffsets(frame_nbytes=case['frame_nbytes'])
    for step in case['steps']:
        frame_nr, position = step['set']
        table[frame_nr] = position
        assert table.frame_nr == step['frame_nr']
        assert table.offset == step['offset']
        assert [table[k] for k in range(32)] == step['lookup']
        assert len(table) == len(step['frame_nr'])
    assert repr(table) == case['repr']


@pytest.mark.parametrize('trial', range(len(GOLD['fuzz'])))
def test_from_index_gives_the_minimal_table(trial):
    """A dense index of positions folds into the table that assigning every
    frame in turn would leave -- and that is the reference's final state when
    its look-ups are assigned frame by frame."""
    case = GOLD['fuzz'][trial]
    lookup = case['steps'][-1]['lookup']
    table = RawOffsets.from_index(lookup, case['frame_nbytes'], known=np.ones(32, bool))
    assert [table[k] for k in range(32)] == lookup
    one_by_one = RawOffsets(frame_nbytes=case['frame_nbytes'])
    for k, position in enumerate(lookup):
        one_by_one[k] = position
    assert (table.frame_nr, table.offset) == (one_by_one.frame_nr, one_by_one.offset)
    # missing frames tell nothing
    known = np.ones(32, bool)
    known[1::3] = False
    sparse = RawOffsets.from_index(lookup, case['frame_nbytes'], known=known)
    assert all(sparse[k] == lookup[k] for k in range(32) if known[k])
    assert len(sparse) <= len(table)


@pytest.mark.parametrize('frame_nbytes', [0, 10, 100])
def test_without_entries(frame_nbytes):
    table = RawOffsets(frame_nbytes=frame_nbytes)
    assert [table[k] for k in (0, 1, 10)] == [0, frame_nbytes, 10 * frame_nbytes]
    assert len(table) == 0
    assert repr(table) == 'RawOffsets(frame_nr=[], offset=[], frame_nbytes={})'.format(frame_nbytes)


def test_given_lists_and_bad_arguments():
    assert RawOffsets([10], [5])[11] == 5
    table = RawOffsets([5, 15], [-1, 1])
    assert (len(table), table[4], table[11], table[15]) == (2, 0, -1, 1)
    for bad in (1.5, (4,)):
        with pytest.raises(TypeError):
            RawOffsets(frame_nbytes=bad)
    with pytest.raises(ValueError):
        RawOffsets([1], None)
    with pytest.raises(ValueError):
        RawOffsets([5, 15], [0])
