"""SURVEY 8(e) with REAL processes on the one GPU of this box (VERDICT r4 next
6): two fresh interpreters, a gloo process group for the index broadcast
(RCCL refuses two ranks on one device), each rank on cuda:0 with its own
output arena, each decoding its time slab; the parent stitches the slabs and
compares them with the reference-generated goldens.  What this exercises that
the in-process loops of test_parallel_gpu.py do not: process groups, ranks
taken from the group, the collective moving the index between processes,
per-process arenas on a shared device, the slab arithmetic end to end.

No scaling figure comes out of this (one GPU): it is a correctness test."""
import json
import os

import numpy as np
import pytest

import conftest
from conftest import ROOT, load_expected, bits_equal

pytestmark = pytest.mark.gpu

CASES = ['vdif_cfg3_small', 'm5b_c16_b2', 'dada_p2_c4_cplx']


@pytest.mark.parametrize('world', [2, 3])
def test_ranks_in_separate_processes_tile_the_stream(tmp_path, world):
    launcher = conftest.rank_launcher()
    if launcher is None:
        pytest.skip("the rank launcher was not started (no /dev/kfd when the session began)")
    req = {"world": world, "script": os.path.join(ROOT, 'tests', 'mp_rank.py'),
           "args": [str(tmp_path)] + CASES, "timeout": 420}
    launcher.stdin.write(json.dumps(req) + '\n')
    launcher.stdin.flush()
    ans = json.loads(launcher.stdout.readline())
    assert "error" not in ans, ans
    assert ans["rcs"] == [0] * world, "\n----\n".join(ans["tails"])
    reports = [json.load(open(tmp_path / 'report_{}.json'.format(r))) for r in range(world)]
    assert len({r["pid"] for r in reports}) == world and os.getpid() not in {r["pid"] for r in reports}
    assert all(r["backend"] == 'gloo' and r["world"] == world for r in reports)
    assert all(r["arena_block"] and r["arena_sum_ok"] for r in reports)
    for name in CASES:
        exp = load_expected(name)
        edges = [(r["cases"][name]["first"], r["cases"][name]["last"]) for r in reports]
        assert edges[0][0] == 0 and edges[-1][1] == exp.shape[0]
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        assert all(b > a for a, b in edges)
        parts = [np.load(tmp_path / '{}_{}.npy'.format(name, r)) for r in range(world)]
        got = np.concatenate(parts)
        assert bits_equal(got.reshape(exp.shape), exp), name
    assert 'broadcast' in reports[1]["cases"]['vdif_cfg3_small']["how"]
