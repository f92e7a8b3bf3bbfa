"""The replay machinery of the recorded cases checks what it claims to check: a changed
digest, a missing or different exception, a lost warning, an instant 2 ns off, a float a
part in a million off are all reported; instants within 1 ns, floats within 1e-8 and
exceptions private to the two packages with one builtin ancestor pass."""
import numpy as np

import casekit


def _steps(n):
    return [{'op': 'get', 'of': 'x%d' % k} for k in range(n)]


def test_compare_reports_what_differs():
    sha_a, sha_b = 'a' * 64, 'b' * 64
    want = [{'v': {'shape': [4, 8], 'dtype': 'float32', 'sha': sha_a, 'head': [1.0]}},
            {'raises': 'HeaderNotFoundError', 'builtin': 'LookupError', 'msg': 'x'},
            {'v': None, 'warns': [['UserWarning', 'partial buffer']]},
            {'v': {'t': '2014-06-16T05:56:07.000000375'}},
            {'v': 0.0025},
            {'v': {'header': 'VDIFHeader3', 'words': [1, 2, 3]}}]
    same = [{'v': {'shape': [4, 8], 'dtype': 'float32', 'sha': sha_a, 'head': [9.0]}},      # (head is for reading, not compared)
            {'raises': 'HeaderNotFound', 'builtin': 'LookupError', 'msg': 'other words'},
            {'v': None, 'warns': [['UserWarning', 'worded differently']]},
            {'v': {'t': '2014-06-16T05:56:07.000000376'}},
            {'v': 0.0025 * (1 + 5e-9)},
            {'v': {'header': 'VDIFHeader3', 'words': [1, 2, 3]}}]
    assert casekit.compare(_steps(6), want, same) == []
    bad = [{'v': {'shape': [4, 8], 'dtype': 'float32', 'sha': sha_b}},
           {'v': 3},
           {'v': None},
           {'v': {'t': '2014-06-16T05:56:07.000000377'}},
           {'v': 0.0025 * (1 + 1e-6)},
           {'v': {'header': 'VDIFHeader3', 'words': [1, 2, 4]}}]
    diffs = casekit.compare(_steps(6), want, bad)
    assert len(diffs) == 6 and all(('#%d ' % k) in d for k, d in enumerate(diffs)), diffs
    # another builtin ancestor is another exception; an unexpected one is reported with its text
    diffs = casekit.compare(_steps(2), [want[1], {'v': 1}],
                            [{'raises': 'ValueError', 'builtin': 'ValueError', 'msg': ''},
                             {'raises': 'TypeError', 'builtin': 'TypeError', 'msg': 'boom'}])
    assert len(diffs) == 2 and 'boom' in diffs[1]
    # fewer outcomes than recorded
    assert casekit.compare(_steps(2), want[:2], same[:1])[0].startswith('1 outcomes')


def test_runner_reduces_results_to_plain_data(tmp_path):
    class Plain(casekit.Universe):
        def module(self, name):
            raise KeyError(name)

        def package_dirs(self):
            return []
    r = casekit.Runner(Plain(), tmp_path)
    out = r.run([{'op': 'let', 'as': 'a', 'to': {'$zeros': [3, 2], 'dt': 'f4'}, 'quiet': True},
                 {'op': 'get', 'of': 'a'},
                 {'op': 'item', 'of': 'a', 'key': {'$tuple': [{'$slice': [0, 2]}, 1]}},
                 {'op': 'fn', 'name': 'write_file', 'args': [{'$tmp': 'f.bin'}, [{'$hex': '0102'}, {'$fill': [255, 3]}]]},
                 {'op': 'digest', 'path': {'$tmp': 'f.bin'}},
                 {'op': 'get', 'of': 'nothing_of_that_name'},
                 {'op': 'let', 'as': 'r', 'to': {'$rng': 7, 'shape': [5], 'levels': [-1.0, 1.0]}, 'quiet': True},
                 {'op': 'get', 'of': 'r'}])
    r.finish()
    assert out[0] == {'v': None}
    assert out[1]['v']['shape'] == [3, 2] and out[1]['v']['dtype'] == 'float32' and len(out[1]['v']['sha']) == 64
    assert out[2]['v']['shape'] == [2]
    assert out[4]['v'] == {'size': 5, 'sha': casekit.sha256(bytes([1, 2, 255, 255, 255]))}
    assert out[5]['raises'] == 'Missing' and out[5]['builtin'] == 'KeyError'
    again = casekit.Runner(Plain(), tmp_path).run([{'op': 'let', 'as': 'r', 'to': {'$rng': 7, 'shape': [5], 'levels': [-1.0, 1.0]}},
                                                   {'op': 'get', 'of': 'r'}])
    assert again[1] == out[7] and set(np.unique(out[7]['v']['head'])) <= {-1.0, 1.0}      # the same numbers under any numpy


def test_the_two_looser_comparisons_are_no_looser_than_they_say():
    """``some_warns`` compares whether there were warnings; ``we_may_manage`` accepts a result where
    the reference raised -- and only that: another exception, or an exception where the reference
    had none, is still reported."""
    warned = {'v': 1, 'warns': [['UserWarning', 'a'], ['UserWarning', 'b']]}
    st = [{'op': 'call', 'fn': 'f.read', 'some_warns': True}]
    assert casekit.compare(st, [warned], [{'v': 1, 'warns': [['UserWarning', 'one only']]}]) == []
    assert len(casekit.compare(st, [warned], [{'v': 1}])) == 1
    assert len(casekit.compare(st, [{'v': 1}], [warned])) == 1
    gave_up = {'raises': 'AssertionError', 'builtin': 'AssertionError', 'msg': 'Cannot find header nearby.'}
    st = [{'op': 'call', 'fn': 'f.read', 'we_may_manage': True, 'some_warns': True}]
    assert casekit.compare(st, [gave_up], [warned]) == []
    assert casekit.compare(st, [gave_up], [dict(gave_up)]) == []
    assert len(casekit.compare(st, [gave_up], [{'raises': 'OSError', 'builtin': 'OSError', 'msg': ''}])) == 1
    assert len(casekit.compare(st, [{'v': 1}], [gave_up])) == 1
    assert len(casekit.compare(st, [{'v': 1}], [{'v': 2}])) == 1


def test_complex_arrays_travel_as_pairs(tmp_path):
    class Plain(casekit.Universe):
        def module(self, name):
            raise KeyError(name)

        def package_dirs(self):
            return []
    r = casekit.Runner(Plain(), tmp_path)
    out = r.run([{'op': 'let', 'as': 'c', 'to': {'$array': [[[1.0, 2.0], [3.0, -4.0]]], 'dt': 'c8'}},
                 {'op': 'get', 'of': 'c'}])
    r.finish()
    assert out[1]['v']['shape'] == [1, 2] and out[1]['v']['dtype'] == 'complex64'
