"""bench.py's multi-rank plumbing on the CPU: ``python bench.py --gpus 2
--dry-run`` must start two ranks itself (no launcher, no WORLD_SIZE in the
environment), rendezvous over gloo on 127.0.0.1, broadcast the frame index and
print ONE JSON line with ``n_gpus == 2`` from rank 0.  Nothing is decoded in
this mode (no GPU here); the line says so (``dry_run``, ``value`` null).

The line is COMPACT (the driver keeps the tail of stdout
only and could not parse a 21 KB line): the contract's keys, `roofline`,
`cpu_baseline`, `checks_ok`, at most 2,000 bytes, the LAST line of stdout; the
full record of the legs is the detail file (``--detail``)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "checks_ok",
                 "detail")


def _run(argv, env_extra=None, detail=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    if detail is not None:
        argv = argv + ['--detail', str(detail)]
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, lines


def _check_compact(stdout):
    """What the driver needs of stdout: its LAST line is one JSON object of at
    most 2,000 bytes with every contract key."""
    last = stdout.rstrip('\n').splitlines()[-1]
    assert len(last.encode()) <= 2000, len(last)
    line = json.loads(last)
    for k in CONTRACT_KEYS:
        assert k in line, k
    return line


def test_gpus_2_spawns_two_ranks_and_prints_one_line(tmp_path):
    detail = tmp_path / 'd.json'
    r, lines = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'], detail=detail)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    line = _check_compact(r.stdout)
    assert line == json.loads(lines[0]) and line['detail'] in (str(detail), os.path.relpath(str(detail), ROOT))
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2
    assert line['dry_run'] is True and line['value'] is None
    # of the legs only the sharded one shows in an N > 1 line, cut down to what proves the ranks took part
    assert set(line['cfg3']) == {'collective', 'rank0_footprint_GiB'} and line['cfg3']['collective']['ranks_seen'] == 2
    line = json.loads(detail.read_text())
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2
    assert line['dry_run'] is True and line['value'] is None
    assert line['steps'] == 3 and line['warmup'] == 1
    assert line['scaling'] == 'weak' and line['dtype'] == 'float32'
    coll = line['cfg3']['collective']
    assert coll['ranks_seen'] == 2 and coll['bytes'] == 2 * 1000 * 8 * 8
    assert line['cfg3']['index_ok'] is True
    assert line['slab_of_rank0'] == [0, 1000]
    assert abs(line['max_over_ranks_s'] - 0.002) < 1e-9         # rank 1's value: the MAX was taken


def test_single_rank_line_and_world_mismatch(tmp_path):
    r, lines = _run(['--dry-run'], detail=tmp_path / 'd.json')
    assert r.returncode == 0 and len(lines) == 1
    assert _check_compact(r.stdout)['n_gpus'] == 1
    # under a launcher the ranks come from the environment; a mismatch with
    # --gpus is an error, not a silently different run
    r, lines = _run(['--gpus', '4'], {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and not lines
    assert 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_under_torch_distributed_run_like_the_driver(tmp_path):
    """The driver's way for N > 1: ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    ...`` -- ranks from the environment, no self-spawn.  Rank 0 picks up the CPU
    baseline a parent handed over (BB_BENCH_CPU_BASELINE_JSON) and reports
    `traffic` null with the reason."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    handed = tmp_path / 'cpu.json'
    handed.write_text(json.dumps({"value": 123.5, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                  "sample": "test"}))
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['BB_BENCH_CPU_BASELINE_JSON'] = str(handed)
    env.setdefault('OMP_NUM_THREADS', '1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--dry-run', '--detail', str(tmp_path / 'd.json')],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = _check_compact(r.stdout)
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['dry_run'] is True
    assert line['cpu_baseline']['value'] == 123.5 and line['cpu_baseline']['kind'] == 'port'
    assert line['roofline']['traffic'] is None
    line = json.loads((tmp_path / 'd.json').read_text())
    assert 'N = 1' in line['roofline']['traffic_detail']['reason']
    assert line['cfg3']['collective']['ranks_seen'] == 2 and line['cfg3']['index_ok'] is True


def test_world_8_exactly_as_the_driver_starts_it(tmp_path):
    """No 8-GPU node has run this bench yet (SCALE records are skips): the N = 8 path is
    rehearsed here as the driver would start it -- ``torch.distributed.run --nproc-per-node 8``
    -- over gloo.  ONE line under 2,000 bytes; all eight ranks seen by the index broadcast;
    one per-rank figure each; and what rank 0 must hold (the whole 64 GiB cfg3 file, its scan
    records and the index) stated in the line."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1',
                        '--dry-run', '--detail', str(tmp_path / 'd8.json')],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    assert len(lines[0].encode()) < 2000
    line = _check_compact(r.stdout)
    assert line['n_gpus'] == 8 and line['ranks_seen'] == 8 and line['scaling'] == 'weak'
    assert line['cfg3']['collective']['ranks_seen'] == 8
    assert line['cfg3']['collective']['bytes'] == 8 * 1000 * 8 * 8
    assert len(line['per_rank']['kernel_ms_avg']) == 8
    assert line['per_rank']['kernel_ms_avg'] == [float(k + 1) for k in range(8)]       # every rank's own value
    # 8 slabs of 8 GiB (whole frame sets) + 24 B per frame of records and index
    assert 64.0 < line['cfg3']['rank0_footprint_GiB'] < 64.5
    detail = json.loads((tmp_path / 'd8.json').read_text())
    assert detail['slab_of_rank0'] == [0, 1000] and detail['cfg3']['index_ok'] is True
    assert abs(detail['max_over_ranks_s'] - 0.008) < 1e-9


def test_a_false_self_check_fails_the_bench(tmp_path):
    """every self-check of the line is folded into
    `checks_ok`, and a false one makes bench.py exit non-zero (the line is
    still printed).  One rank and two ranks (the launcher must pass rank 0's
    status on)."""
    detail = tmp_path / 'd.json'
    r, lines = _run(['--dry-run'], detail=detail)
    assert r.returncode == 0
    line = json.loads(lines[0])
    assert line['checks_ok'] is True and 'failed_checks' not in line
    assert json.loads(detail.read_text())['checks']['cfg3.index_ok'] is True
    r, lines = _run(['--dry-run'], {'BB_BENCH_FORCE_CHECK_FALSE': 'cfg3.index_ok'}, detail=detail)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    line = _check_compact(r.stdout)
    assert line['checks_ok'] is False and line['failed_checks'] == ['cfg3.index_ok']
    assert json.loads(detail.read_text())['checks']['cfg3.index_ok'] is False
    assert 'self-checks FAILED' in r.stderr and 'cfg3.index_ok' in r.stderr
    r, lines = _run(['--gpus', '2', '--dry-run'], {'BB_BENCH_FORCE_CHECK_FALSE': 'cfg3.index_ok'}, detail=detail)
    assert r.returncode != 0 and len(lines) == 1
    assert json.loads(lines[0])['checks_ok'] is False


def test_collect_checks_reads_every_leg():
    """The verdict is false when any leg's check is false, when a leg that
    carries a check failed before evaluating it, and true otherwise."""
    sys.path.insert(0, ROOT)
    import bench
    good = {"sanity_spot_check": True, "parity_digests": {"all_match": True},
            "invalid_fill": {"flagged_frame_is_fill": True, "neighbour_frame_is_data": True,
                             "all_frames_invalid": {"output_is_fill": True}},
            "cfg3": {"sanity_spot_check": True}}
    ok, checks = bench.collect_checks(good)
    assert ok and len(checks) == 6 and all(checks.values())
    import copy
    for path in (("sanity_spot_check",), ("parity_digests", "all_match"),
                 ("invalid_fill", "neighbour_frame_is_data"), ("invalid_fill", "flagged_frame_is_fill"),
                 ("invalid_fill", "all_frames_invalid", "output_is_fill"), ("cfg3", "sanity_spot_check")):
        bad = copy.deepcopy(good)
        d = bad
        for k in path[:-1]:
            d = d[k]
        d[path[-1]] = False
        assert bench.collect_checks(bad)[0] is False, path
    bad = dict(good, invalid_fill={"error": "RuntimeError('x')"})
    ok, checks = bench.collect_checks(bad)
    assert ok is False and checks["invalid_fill.neighbour_frame_is_data"] is False
    assert bench.collect_checks({"sanity_spot_check": True})[0] is True


def test_a_full_record_still_makes_a_compact_line():
    """The line the driver could not read in round 4 (21 KB, profiles/r04fin_bench.json)
    through `compact_line`: at most 2,000 bytes, headline / roofline / cpu_baseline
    figures intact, the secondary legs as one short block."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, 'profiles', 'r04fin_bench.json')) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 15000
    c = bench.compact_line(full)
    text = json.dumps(c, separators=(',', ':'))
    assert len(text.encode()) <= bench.LINE_LIMIT == 2000
    for k in CONTRACT_KEYS:
        assert k in c, k
    assert c['value'] == full['value'] and c['ms_per_step'] == full['ms_per_step']
    rf = c['roofline']
    assert rf['frac'] == full['roofline']['frac'] and rf['bound'] == 'hbm' and rf['peak'] == 8000.0
    assert rf['traffic'] == full['roofline']['traffic'] and rf['traffic_over_algorithmic'] == 1.0001
    assert rf['kernel_time_over_write_only_time'] == 1.1108
    assert rf['kernel'].startswith('k_decode_flat_lds')
    cb = c['cpu_baseline']
    assert cb['value'] == 1960.29 and cb['cores'] == 1 and cb['kind'] == 'port'
    assert cb['all_cores'] == {"value": 7644.4, "cores": 128} and cb['reference_as_written_estimate'] == 718.58
    sec = c['secondary']
    assert sec['cfg3'] == 0.857 and sec['mark5b'] == 0.834 and sec['gather_select'] > 0.5
    assert sec['vdif_4bit'] == 0.836 and sec['enc4'] == 0.73 and sec['vdif_8bit'] == 0.835
    assert sec['guppi_tf'] == 0.833 and sec['guppi_tf_pick'] != sec['guppi_tf'] and len(sec) >= 20
    # a line that cannot fit sheds its optional blocks instead of growing
    fat = dict(full, other_configs=full['other_configs'] * 1,
               cpu_baseline=dict(full['cpu_baseline'], sample='x' * 5000),
               config=dict(full['config'], workload='w' * 1500))
    assert len(json.dumps(bench.compact_line(fat), separators=(',', ':')).encode()) <= 2000


def test_physical_core_count_is_stated():
    """the all-cores CPU leg runs one process per PHYSICAL core of
    the affinity mask and says how many that is."""
    sys.path.insert(0, ROOT)
    import bench
    nphys, nlog = bench.physical_cores()
    assert 1 <= nphys <= nlog == len(os.sched_getaffinity(0))
