#!/usr/bin/env python3
"""Headline benchmark: decoded Msamples/s + achieved HBM GB/s, VDIF 2-bit.

One step = one pass of the hot path (header scan -> index build -> packed
sample decode) over one synthetic file image that is already resident in HBM:
BASELINE.json configs[1], "synthetic 8 GiB single-thread VDIF, 2-bit real,
1 channel, EDV 0" (8032-byte frames, 32000 samples per frame).

``python bench.py --gpus N``: one process per GPU.  Under
``python -m torch.distributed.run`` (the driver's way) the ranks come from the
environment; started plainly with N > 1 this process -- before it touches the
GPU -- starts ``torch.distributed.run`` with N workers itself and passes their
output through.  Every rank decodes its own time slab of the same size (weak
scaling, no data-path collective: frames are independent, SURVEY.md section
8e); rank 0 prints ONE JSON line.

Besides the contract's keys the line carries
  roofline      dominant kernel (named by the library: bb_last_kernel) timed
                with HIP events on the launching stream; `traffic` = HBM bytes
                per launch from two rocprofv3 --pmc child passes of this same
                script (run before this process touches the GPU), or the
                committed profiles/traffic_latest.json when that fails
  cpu_baseline  the NumPy restatement of the reference's per-frame loop
                (oracle/, kind "port") on a bounded sample, rank 0 at N = 1,
                with the calibration against the real reference
                (tests/golden/cpu_calibration.json)
  api_read      the same 8 GiB image through the drop-in API:
                ``vdif.open(<device tensor>, 'rs').read(out=out)``
  invalid_fill  the headline step with `invalid_data` set in 1 % of the frames
                (the fill path must cost nothing)
  cfg3          BASELINE configs[2] (8-thread 2-bit complex 16-channel VDIF):
                rank 0 scans the whole file and builds the frame index, ONE
                broadcast (RCCL) replicates it, every rank decodes its slab
  parity_digests sha256 of golden files decoded here vs the reference's digests
  other_configs Mark 5B / Mark 4 / GUPPI / DADA / 8-thread real VDIF kernels
                on inputs that decode to the headline's output size (N = 1 only)
  pipeline      file -> HBM -> decode: open(path).read() of 2 GiB files (VDIF cfg2 /
                cfg3, Mark 5B, Mark 4, GUPPI, DADA) from the page cache, GB/s of
                file bytes against the pinned H2D rate, per-window times
  mid_size      the launch sizes ordinary read() calls issue (2^15, 2^16, 2^18
                cfg2 frames; GUPPI 8 GiB in) into FRESH outputs, torch.empty
                against the placement arena the readers allocate from, min /
                median / max over the draws (N = 1 only)
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X spec peak (MI355X_MICROARCH.md)
FRAME_NBYTES = 8032
HEADER_NBYTES = 32
PAYLOAD_NBYTES = 8000
SPF = 32000                     # samples per frame (2-bit, real, 1 channel)
FRAME_RATE = 1000               # frames per second -> 32 MHz sample rate
CFG3_THREADS = 8
CFG3_NCHAN = 16
CFG3_ORDER = (1, 3, 5, 7, 0, 2, 4, 6)       # thread id at disk position p (sample.vdif's order)
CFG3_SET_RATE = 1000


def _s32(x):
    return x - (1 << 32) if x >= (1 << 31) else x


def make_file_image_on_device(nsets, seed, first_set, device, nthread=1, nchan=1,
                              complex_data=False, order=(0,), set_rate=FRAME_RATE, into=None):
    """VDIF file image born in HBM: uniform random payload bytes + EDV-0
    headers (seconds / frame_nr from the frame-set index, thread ids in
    `order`).  Same header words as baseband_amd.synth / the reference writer
    would produce.  `into`: optional uint8 tensor of the right size to fill.
    Returns (uint8 tensor, header0)."""
    from baseband_amd.vdif.header import VDIFHeader
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    words_per_frame = FRAME_NBYTES // 4
    nframes = nsets * nthread
    img = (torch.empty(nframes * words_per_frame, dtype=torch.int32, device=device) if into is None
           else into.view(torch.int32))
    assert img.numel() == nframes * words_per_frame
    step = 1 << 28
    for lo in range(0, img.numel(), step):          # bounded temporaries
        hi = min(img.numel(), lo + step)
        img[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g,
                                   device=device, dtype=torch.int64).to(torch.int32)
    h0 = VDIFHeader.fromvalues(edv=0, bps=2, nchan=nchan, complex_data=complex_data,
                               payload_nbytes=PAYLOAD_NBYTES, station='AA',
                               thread_id=order[0],
                               time=np.datetime64('2020-01-01T00:00:00'))
    w = [int(x) for x in h0.words]
    v = img.view(nsets, nthread, words_per_frame)
    idx = torch.arange(first_set, first_set + nsets, device=device, dtype=torch.int64)[:, None]
    v[:, :, 0] = (w[0] + idx // set_rate).to(torch.int32)
    v[:, :, 1] = ((w[1] & 0xff000000) + idx % set_rate).to(torch.int32)
    v[:, :, 2] = _s32(w[2])
    tid = torch.tensor(list(order), device=device, dtype=torch.int64)[None, :]
    v[:, :, 3] = ((w[3] & ~(0x3ff << 16)) | (tid << 16)).to(torch.int32) if nthread > 1 else _s32(w[3])
    v[:, :, 4:8] = 0
    return img.view(torch.uint8), h0


def empty_with_patience(n, dtype, device, tries=12):
    """``torch.empty`` for the 127.5 GiB output.  The image was allocated just
    before, and an arena that had to try several candidate steps has released up
    to 144 GiB a moment ago: memory the driver is still clearing is not
    allocatable yet (seen with tools/experiments/arena_probe3.cpp), so an out-of-memory here
    is retried for a few seconds before it counts."""
    for k in range(tries):
        try:
            return torch.empty(n, dtype=dtype, device=device)
        except torch.cuda.OutOfMemoryError:
            if k == tries - 1:
                raise
            torch.cuda.empty_cache()
            time.sleep(0.5)


def image_buffer(nbytes, device):
    """Device memory for a file image, allocated the way the package keeps
    file bytes in HBM (`fh.stage()`, the staged copy of a large read):
    `baseband_amd.empty_output(dtype=uint8)` -- arena memory from 1 GiB on,
    torch.empty below or with BB_ARENA=0.  Returns (tensor, "arena" | "torch")."""
    import baseband_amd
    from baseband_amd import arena
    t = baseband_amd.empty_output((int(nbytes),), dtype=torch.uint8, device=device)
    ar = arena.default(device)
    return t, ("arena" if ar is not None and ar.owns(t) else "torch")


# ---------------------------------------------------------------- CPU baseline
def _cpu_worker(args):
    """One process of the all-cores CPU leg: its own slab of cfg2 frames through
    the reference-as-written loop for about `seconds`."""
    seed, nframes, seconds = args
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bb_oracle_np as orc
    from baseband_amd import synth
    image, _ = synth.random_vdif(seed, nframes, payload_nbytes=PAYLOAD_NBYTES,
                                 frame_rate=FRAME_RATE)
    orc.vdif_read(image, frame_rate=FRAME_RATE)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.vdif_read(image, frame_rate=FRAME_RATE)
        reps += 1
    return reps * nframes * SPF, time.perf_counter() - t0


def physical_cores():
    """(physical cores this process may run on, logical CPUs it may run on):
    distinct (physical id, core id) pairs of /proc/cpuinfo among the CPUs of
    the affinity mask.  SURVEY 8(d): the all-cores leg runs N = physical cores
    processes, N stated."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    try:
        cores, cur = {}, {}
        with open('/proc/cpuinfo') as f:
            for ln in f.read().split('\n') + ['']:
                if ':' in ln:
                    k, v = ln.split(':', 1)
                    cur[k.strip()] = v.strip()
                elif cur:
                    if 'processor' in cur:
                        cores[int(cur['processor'])] = (cur.get('physical id', '0'), cur.get('core id', cur['processor']))
                    cur = {}
        phys = {cores[c] for c in allowed if c in cores}
        if phys:
            return len(phys), len(allowed)
    except Exception:
        pass
    return len(allowed), len(allowed)


def cpu_baseline(target_seconds=12.0):
    """Reference-as-written loop (NumPy port) on a bounded sample: one core (how
    the reference runs), all host cores over disjoint frame slabs (the
    pickle-to-processes advice of the reference's performance tips), and the
    bare LUT `take` without the per-frame loop as the NumPy ceiling.  Called
    before anything touches the GPU, so that forking workers is safe."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import bb_oracle_np as orc
    from baseband_amd import synth
    nframes = 4000
    image, _ = synth.random_vdif(12345, nframes, payload_nbytes=PAYLOAD_NBYTES,
                                 frame_rate=FRAME_RATE)
    orc.vdif_read(image, frame_rate=FRAME_RATE)           # warm (LUT, page faults)
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.vdif_read(image, frame_rate=FRAME_RATE)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or reps >= 2000:
            break
    msps = reps * nframes * SPF / dt / 1e6
    result = {"value": round(msps, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
              "sample": "{} x {} frames of the same cfg2 layout ({:.1f} MiB each), "
                        "oracle/bb_oracle_np.vdif_read (per-frame NumPy LUT take loop)"
                        .format(reps, nframes, image.size / 2 ** 20),
              "host": "{} physical cores / {} logical CPUs; numpy {}".format(*physical_cores(), np.__version__)}
    # how the port relates to the real reference (measured in the development
    # container, where the reference can be imported: tools/calibrate_cpu_baseline.py)
    try:
        with open(os.path.join(ROOT, 'tests', 'golden', 'cpu_calibration.json')) as f:
            cal = json.load(f)
        ratio = float(cal["ratio_port_over_reference"])
        qlo, qhi = cal.get("ratio_port_over_reference_quartile_range", [ratio, ratio])
        result["calibration"] = {
            "ratio_port_over_reference": ratio,
            "ratio_from": cal.get("ratio_from", "medians"),
            "ratio_quartile_range": [qlo, qhi],
            "ratio_port_over_reference_verify_false": cal.get("ratio_port_over_reference_verify_false"),
            "reference_as_written_estimate_Msps": round(msps / ratio, 2),
            "reference_as_written_estimate_range_Msps": [round(msps / qhi, 2), round(msps / qlo, 2)],
            "what": "the port's figure divided by the MEDIAN ratio reference / port measured where the "
                    "reference can be imported (one pinned core, 25 interleaved rounds, medians and quartiles)",
            "measured_on": cal.get("host"),
            "reference_Msps_there": cal["reference"]["verify_true_Msps"],
            "port_Msps_there": cal["port"]["Msps"],
            "source": "tests/golden/cpu_calibration.json (tools/calibrate_cpu_baseline.py: real "
                      "baseband.vdif.open().read() vs the port on the same seeded file, outputs bit-identical)"}
    except Exception as exc:
        result["calibration"] = {"error": repr(exc)}
    # bare take: every payload byte of the sample through the 256 x 4 table in
    # one call (no headers, no per-frame Python)
    try:
        payload = np.ascontiguousarray(
            image.reshape(nframes, FRAME_NBYTES)[:, HEADER_NBYTES:]).reshape(-1)
        lut = orc.byte_lut('vdif', 2)
        dest = np.empty((payload.size, 4), np.float32)
        step = PAYLOAD_NBYTES                              # cache-sized pieces are fastest
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            for i in range(0, payload.size, step):
                np.take(lut, payload[i:i + step], axis=0, out=dest[i:i + step], mode='clip')
            n += 1
        result["bare_take"] = {"value": round(n * payload.size * 4 / (time.perf_counter() - t0) / 1e6, 1),
                               "unit": "Msamples/s", "cores": 1,
                               "what": "np.take(lut, payload, out=preallocated) in payload-sized "
                                       "pieces: no headers, no index, no allocation"}
    except Exception as exc:                              # report, never fail the bench
        result["bare_take"] = {"error": repr(exc)}
    # all cores: one forked worker per core, 500-frame slabs (64 MiB of output each)
    try:
        import multiprocessing as mp
        nphys, nlogical = physical_cores()
        nproc = max(1, nphys)
        with mp.get_context('fork').Pool(nproc) as pool:
            parts = pool.map(_cpu_worker, [(1000 + i, 500, 5.0) for i in range(nproc)])
        total = sum(p[0] for p in parts)
        slowest = max(p[1] for p in parts)
        result["all_cores"] = {"value": round(total / slowest / 1e6, 1), "unit": "Msamples/s",
                               "cores": nproc, "physical_cores": nphys, "logical_cpus": nlogical,
                               "what": "{} processes (one per physical core of {} logical CPUs) x 500-frame slabs "
                                       "for 5 s each, same loop".format(nproc, nlogical)}
    except Exception as exc:
        result["all_cores"] = {"error": repr(exc)}
    return result


# ------------------------------------------------------------------ HBM traffic
def _git_commit():
    try:
        return subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'],
                              capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def _run_group(cmd, cwd, env, timeout):
    """subprocess.run with the child in its own process group, which is killed
    as a whole on timeout: a profiler that stops answering must not leave the
    program it started on the GPU behind (this process is about to allocate
    nearly all of HBM).  Returns an object with returncode / stdout / stderr."""
    import signal
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        out, err = p.communicate()
        raise RuntimeError("timed out after {} s: {}".format(timeout, ' '.join(cmd[:4])))
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def live_traffic(gib, timeout=150):
    """HBM bytes per launch of the decode kernel from the memory-side counters:
    two child runs of THIS script under ``rocprofv3 --pmc`` (FETCH_SIZE and
    WRITE_SIZE cannot share a pass; MI355X_MICROARCH.md, rocprofv3 PMC slots),
    program directly after ``--``.  Called before this process touches the
    GPU.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes,
    same guide, HBM section).  Returns a dict or raises."""
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix='bbpmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    vals = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '-d', d, '-o', 'c', '--output-format', 'csv', '--',
                   sys.executable, os.path.join(ROOT, 'bench.py'), '--pmc-child',
                   '--steps', '1', '--warmup', '1', '--gib', repr(gib)]
            r = _run_group(cmd, cwd='/tmp', env=env, timeout=timeout)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError("rocprofv3 --pmc {} failed (rc {}): {}".format(
                    counter, r.returncode, (r.stderr or '')[-300:]))
            rows = []
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if 'k_decode' in row['Kernel_Name'] and row['Counter_Name'] == counter:
                        rows.append((int(float(row.get('Grid_Size') or 0)), float(row['Counter_Value'])))
            if not rows:
                raise RuntimeError("no k_decode rows for " + counter)
            # the headline launches are the ones with the largest grid (the output
            # arena probes a new step with short launches of the same kernel)
            top = max(g for g, _ in rows)
            got = [v for g, v in rows if g == top]
            vals[counter] = sum(got) / len(got) * 1024.0            # counters are in KiB
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run",
            "fetch_bytes_raw": vals['FETCH_SIZE'], "write_bytes": vals['WRITE_SIZE'],
            "fetch_bytes_corrected": 2 * vals['FETCH_SIZE'],
            "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
            "hbm_bytes_per_launch": 2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE'],
            "commit": _git_commit(), "date": time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())}


def file_traffic():
    with open(os.path.join(ROOT, 'profiles', 'traffic_latest.json')) as f:
        d = json.load(f)
    d["source"] = "committed file (not from this run): " + str(d.get("source"))
    return d


# --------------------------------------------------------------------- helpers
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


CPU_JSON_ENV = 'BB_BENCH_CPU_BASELINE_JSON'


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start N workers with
    torch.distributed.run (this process has not touched the GPU) and pass
    their output through.  The CPU baseline is timed HERE, before the workers
    exist (the host is otherwise idle, no rank waits for it), and handed to
    rank 0 as a file so that the N > 1 line carries `cpu_baseline` too."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    tmp = None
    if not args.no_cpu_baseline and not args.dry_run and os.path.exists('/dev/kfd'):
        try:
            cpu = cpu_baseline()
            cpu["timed_by"] = "the parent of the {} ranks, before they were started".format(args.gpus)
            fd, tmp = tempfile.mkstemp(prefix='bb_cpu_', suffix='.json', dir='/tmp')
            with os.fdopen(fd, 'w') as f:
                json.dump(cpu, f)
            env[CPU_JSON_ENV] = tmp
        except Exception as exc:                    # the bench goes on without it
            print("bench.py: cpu_baseline failed in the parent: {!r}".format(exc), file=sys.stderr)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py')] + argv
    try:
        return subprocess.run(cmd, env=env).returncode
    finally:
        if tmp:
            try:
                os.remove(tmp)
            except OSError:
                pass


FORCE_FAIL_ENV = 'BB_BENCH_FORCE_CHECK_FALSE'       # tests: make the named check read false
CHECKS_RC = 3                                       # exit status when a check of the line is false


def collect_checks(line):
    """Every self-check the line carries, folded into one verdict.  Returns
    (checks_ok, {name: bool}).  A leg that is in the line but failed before
    its check was evaluated counts as false: a check that did not run is not a
    passed check.  ``BB_BENCH_FORCE_CHECK_FALSE=<name>`` forces one false
    (tests/test_bench_cli.py: the exit status must follow)."""
    checks = {}

    def put(name, leg, *path):
        v = leg
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        checks[name] = v is True

    put("headline.sanity_spot_check", line, "sanity_spot_check")
    if "parity_digests" in line:
        put("parity_digests.all_match", line, "parity_digests", "all_match")
    if "invalid_fill" in line:
        put("invalid_fill.flagged_frame_is_fill", line, "invalid_fill", "flagged_frame_is_fill")
        put("invalid_fill.neighbour_frame_is_data", line, "invalid_fill", "neighbour_frame_is_data")
        put("invalid_fill.all_invalid_output_is_fill", line, "invalid_fill", "all_frames_invalid", "output_is_fill")
    if "cfg3" in line:
        if line.get("dry_run"):
            put("cfg3.index_ok", line, "cfg3", "index_ok")
        else:
            put("cfg3.sanity_spot_check", line, "cfg3", "sanity_spot_check")
            if isinstance(line["cfg3"], dict) and "rank_local_scan" in line["cfg3"]:
                put("cfg3.rank_local_scan.index_equals_broadcast", line, "cfg3", "rank_local_scan",
                    "index_equals_broadcast")
    if "pipeline" in line and not (isinstance(line["pipeline"], dict) and "skipped" in line["pipeline"]):
        put("pipeline.all_match", line, "pipeline", "all_match")
    if "other_configs" in line:
        oc = line["other_configs"]
        checks["other_configs.spot_checks"] = bool(oc) and all(
            isinstance(c, dict) and "error" not in c and c.get("spot_check", True) is True for c in oc)
    forced = os.environ.get(FORCE_FAIL_ENV)
    if forced:
        checks[forced] = False
    return all(checks.values()), checks


def finish(line):
    """Attach `checks_ok` / `checks`, print THE line, return the exit status."""
    ok, checks = collect_checks(line)
    line["checks_ok"] = ok
    line["checks"] = checks
    print(json.dumps(line), flush=True)
    if not ok:
        print("bench.py: self-checks FAILED: " + ", ".join(k for k, v in checks.items() if not v),
              file=sys.stderr, flush=True)
        return CHECKS_RC
    return 0


def timed_launches(fn, reps):
    """Median and mean ms of `fn` (one launch) by HIP events on torch's current
    stream, which is the stream the library launches on (kernels._stream)."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.mean(ts))


def expand_2bit(raw, lev):
    """Host re-expansion of 2-bit VDIF payload bytes from the library's own
    level table (4 samples per byte, least significant pair first): the
    in-bench sanity spot check, NOT the parity proof (that lives in tests/)."""
    return lev[(raw[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3].reshape(-1)


# ------------------------------------------------------------------ dry run
def dry_run(args, rank, world):
    """CPU rehearsal of the multi-rank plumbing (tests/test_bench_cli.py): gloo
    rendezvous, slab partition, the index broadcast, barrier + max-over-ranks
    timing, one JSON line from rank 0.  Nothing is decoded and nothing is
    measured: ``value`` is null and ``dry_run`` is true."""
    import torch.distributed as dist
    from baseband_amd.parallel import frame_slab, broadcast_frame_index
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    nsets = 1000 * world
    lo, hi = frame_slab(nsets, rank, world)
    src = torch.arange(nsets * CFG3_THREADS, dtype=torch.int64) * FRAME_NBYTES + HEADER_NBYTES \
        if rank == 0 else None
    t0 = time.perf_counter()
    if world > 1:
        src = broadcast_frame_index(src, nsets * CFG3_THREADS, src_rank=0)
    coll_ms = (time.perf_counter() - t0) * 1e3
    ok = bool((src[lo * CFG3_THREADS:hi * CFG3_THREADS]
               == torch.arange(lo * CFG3_THREADS, hi * CFG3_THREADS) * FRAME_NBYTES + HEADER_NBYTES).all())
    seen = torch.ones(1)
    elapsed = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(seen)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        cpu = None
        handed = os.environ.get(CPU_JSON_ENV)
        if handed and os.path.exists(handed):       # what the parent of an N > 1 run timed
            with open(handed) as f:
                cpu = json.load(f)
        rc = finish({
            "metric": "decoded Msamples/s, VDIF 2-bit 1-thread (scan + index + decode, input resident in HBM)",
            "sanity_spot_check": True,              # nothing is decoded in a dry run
            "cpu_baseline": cpu,
            "roofline": {"traffic": None, "traffic_detail": {
                "hbm_bytes_per_launch": None, "reason": "dry run" if world == 1 else "counter passes run at N = 1 only"}},
            "value": None, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "float32", "data": "synthetic",
            "dry_run": True, "ranks_seen": int(seen.item()), "slab_of_rank0": [lo, hi],
            "max_over_ranks_s": float(elapsed.item()),
            "cfg3": {"collective": {"bytes": nsets * CFG3_THREADS * 8, "ms": round(coll_ms, 3),
                                    "ranks_seen": int(seen.item()), "backend": "gloo"},
                     "index_ok": ok}})
    else:
        rc = 0
    if world > 1:
        dist.destroy_process_group()
    return rc


def parity_digests():
    """Bit-exactness verdict per configuration:
    the small reference-written golden files of every format are decoded
    through the drop-in API on this GPU and the sha256 of the decoded array is
    compared with the digest of the REFERENCE's output committed in
    tests/golden/manifest.json (written by oracle/gen_golden.py from the real
    reference).  Outside every timed region; the parity proof proper is
    tests/ (-m gpu)."""
    import hashlib
    import baseband_amd as bb
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest.json')) as f:
        cases = json.load(f)['cases']
    plan = [('sample_vdif', bb.vdif.open, {}),
            ('vdif_cfg2_small', bb.vdif.open, None), ('vdif_cfg3_small', bb.vdif.open, None),
            ('m5b_c16_b2', bb.mark5b.open, 'm5b'), ('m4_t64_f4', bb.mark4.open, 'm4'),
            ('guppi_cf_c64_ov0', bb.guppi.open, {}), ('dada_p2_c4_cplx', bb.dada.open, {})]
    res = {}
    for name, opener, kw in plan:
        c = cases[name]
        if kw is None:
            kw = dict(sample_rate=c['frame_rate'] * c['samples_per_frame'])
        elif kw == 'm5b':
            kw = dict(sample_rate=c['frame_rate'] * c['samples_per_frame'], kday=c['kday'],
                      nchan=c['nchan'], bps=c['bps'])
        elif kw == 'm4':
            kw = dict(sample_rate=c['frame_rate'] * c['samples_per_frame'], ntrack=c['ntrack'],
                      decade=2010, verify=False)
        try:
            with opener(os.path.join(ROOT, 'tests', 'golden', c['file']), 'rs', squeeze=False, **kw) as fh:
                got = fh.read().cpu().numpy()
            digest = hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()
            res[name] = {"shape": list(got.shape), "sha256_matches_reference": digest == c['sha256']}
        except Exception as exc:
            res[name] = {"error": repr(exc)[:200]}
    res["all_match"] = all(v.get("sha256_matches_reference") is True for v in res.values())
    return res


# --------------------------------------------------------------------- legs
def leg_cfg3(args, rank, world, device, dist, out):
    """BASELINE configs[2]: 8-thread 2-bit complex 16-channel VDIF sharded by
    time slab.  Rank 0 holds the whole file image, scans every header and
    builds the dense (frame set, thread) -> payload offset index; ONE
    broadcast replicates it (RCCL over xGMI under 'nccl'); every rank rebases
    its slab of the index and decodes its own bytes into its own HBM."""
    from baseband_amd import kernels, _lib
    from baseband_amd.parallel import frame_slab, broadcast_frame_index, local_index
    set_nbytes = FRAME_NBYTES * CFG3_THREADS
    nsets = int(args.cfg3_gib * 2 ** 30) // set_nbytes
    nsets_world = nsets * world
    lo, hi = frame_slab(nsets_world, rank, world)
    kw = dict(nthread=CFG3_THREADS, nchan=CFG3_NCHAN, complex_data=True, order=CFG3_ORDER,
              set_rate=CFG3_SET_RATE)
    if rank == 0:
        # the scanning rank holds the whole file (one allocation, filled slab by
        # slab: no concatenation copy next to the 127.5 GiB output buffer)
        sb = nsets * set_nbytes
        # footprint of rank 0 in this leg: the whole file (world x slab) + scan
        # records (16 B per frame) + the index (8 B per frame) + random-fill
        # temporaries (4 GiB at most), next to `out`, which the caller holds.
        # Checked against what the driver reports free, BEFORE allocating: at
        # N = 8 this is 64 GiB beside the 127.5 GiB output
        need = world * sb + nsets_world * CFG3_THREADS * 24 + (4 << 30)
        free_b, total_b = torch.cuda.mem_get_info(device)
        free_b += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        if need > free_b:
            raise RuntimeError("cfg3 leg: rank 0 needs {:.1f} GiB (whole file of {} ranks + index) but {:.1f} GiB "
                               "are free: lower --cfg3-gib".format(need / 2 ** 30, world, free_b / 2 ** 30))
        whole = torch.empty(world * sb, dtype=torch.uint8, device=device)
        for r in range(world):
            _, h0 = make_file_image_on_device(nsets, 777 + r, frame_slab(nsets_world, r, world)[0], device,
                                              into=whole[r * sb:(r + 1) * sb], **kw)
        slab = whole[:sb]
        h0 = make_file_image_on_device(1, 777, 0, torch.device('cpu'), **kw)[1]     # header of set 0
    else:
        slab, h0 = make_file_image_on_device(nsets, 777 + rank, lo, device, **kw)
    pattern, mask = h0.invariant_pattern()
    thread_slot = kernels.thread_slot_map(list(range(CFG3_THREADS)), device)
    chunk = CFG3_NCHAN * 2
    nelem = nsets * CFG3_THREADS * PAYLOAD_NBYTES * 4
    o = out[:nelem]
    nentries = nsets_world * CFG3_THREADS
    coll = []
    dec = []
    kname = [None]

    def step(k=None):
        src = None
        if rank == 0:
            recs = kernels.vdif_scan(whole, nsets_world * CFG3_THREADS, FRAME_NBYTES, HEADER_NBYTES,
                                     pattern, mask, h0['seconds'], h0['frame_nr'], CFG3_SET_RATE)
            src = kernels.build_index(recs, nsets_world, CFG3_THREADS, thread_slot)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        if dist is not None:
            src = broadcast_frame_index(src, nentries, src_rank=0, device=device)
        e[1].record()
        local, byte_lo, byte_hi = local_index(src, lo, hi, CFG3_THREADS, PAYLOAD_NBYTES)
        assert byte_lo >= lo * set_nbytes and byte_hi <= hi * set_nbytes
        local = local + (byte_lo - lo * set_nbytes)             # offsets into this rank's slab image
        e[2].record()
        kernels.decode_frames(slab, nsets, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, chunk=chunk,
                              nslot=CFG3_THREADS, src=local, complex_data=True, out=o)
        e[3].record()
        kname[0] = _lib.last_kernel()
        if k is not None:
            coll.append((e[0], e[1]))
            dec.append((e[2], e[3]))

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = torch.ones(1, device=device)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(seen)
    # Second variant: every rank scans ITS OWN slab and
    # builds its own index -- no collective, no serial work on rank 0 -- so that
    # a measured curve separates the broadcast's cost from rank 0's whole-file
    # scan.  (The north_star's form is the one above; this one needs every
    # rank to know where its slab's first frame set lies in time, which a
    # fixed-rate file gives and a file with missing frames does not.)
    last_local = [None]

    def step_local():
        recs = kernels.vdif_scan(slab, nsets * CFG3_THREADS, FRAME_NBYTES, HEADER_NBYTES, pattern, mask,
                                 h0['seconds'], h0['frame_nr'] + lo, CFG3_SET_RATE)
        src_l = kernels.build_index(recs, nsets, CFG3_THREADS, thread_slot)
        kernels.decode_frames(slab, nsets, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, chunk=chunk,
                              nslot=CFG3_THREADS, src=src_l, complex_data=True, out=o)
        last_local[0] = src_l

    step_local()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_local()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed_l = time.perf_counter() - t0
    # the two indices must be the same table
    src_b = None
    if rank == 0:
        src_b = kernels.build_index(
            kernels.vdif_scan(whole, nsets_world * CFG3_THREADS, FRAME_NBYTES, HEADER_NBYTES, pattern, mask,
                              h0['seconds'], h0['frame_nr'], CFG3_SET_RATE), nsets_world, CFG3_THREADS, thread_slot)
    if dist is not None:
        src_b = broadcast_frame_index(src_b, nentries, src_rank=0, device=device)
    loc_b, blo, _ = local_index(src_b, lo, hi, CFG3_THREADS, PAYLOAD_NBYTES)
    same_index = torch.tensor([1.0 if torch.equal(loc_b + (blo - lo * set_nbytes), last_local[0]) else 0.0], device=device)
    if dist is not None:
        t = torch.tensor([elapsed_l], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_l = float(t.item())
        dist.all_reduce(same_index, op=dist.ReduceOp.MIN)
    del src_b, loc_b
    # sanity: set 5 of this slab, every thread, against a host re-expansion
    lev = _lib.get_levels(_lib.CODER_VDIF, 2)
    ok = True
    spf = PAYLOAD_NBYTES * 4 // (2 * CFG3_NCHAN)                 # complex samples per frame
    got = o.view(nsets, spf, CFG3_THREADS, chunk)[5].cpu().numpy()
    for p, t in enumerate(CFG3_ORDER):
        fo = (5 * CFG3_THREADS + p) * FRAME_NBYTES
        raw = slab[fo + HEADER_NBYTES:fo + FRAME_NBYTES].cpu().numpy()
        want = expand_2bit(raw, lev).reshape(spf, chunk)
        ok &= bool(np.array_equal(got[:, t].view(np.uint32), want.view(np.uint32)))
    coll_ms = float(np.mean([a.elapsed_time(b) for a, b in coll]))
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in dec]))
    alg = nsets * CFG3_THREADS * (FRAME_NBYTES + PAYLOAD_NBYTES * 16)
    achieved = alg / (dec_ms * 1e-3) / 1e9
    ncomplex = nsets * CFG3_THREADS * PAYLOAD_NBYTES * 2        # complex samples x threads x channels
    return {
        "workload": "cfg3: synthetic {:.3f} GiB per GPU 8-thread VDIF, 2-bit complex, 16 channels, "
                    "EDV 0, thread order on disk {}".format(nsets * set_nbytes / 2 ** 30, list(CFG3_ORDER)),
        "rank0_file_GiB": round(world * nsets * set_nbytes / 2 ** 30, 3),
        "value": round(ncomplex * world * args.steps / elapsed / 1e6, 1),
        "unit": "M complex samples/s (threads x channels counted)",
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "step": "rank 0: bb_vdif_scan + bb_build_index over the whole file; broadcast; "
                "every rank: rebase (parallel.local_index) + bb_decode_frames of its slab",
        "collective": {"op": "broadcast of the dense frame index", "bytes": nentries * 8,
                       "ms": round(coll_ms, 4), "ranks_seen": int(seen.item()),
                       "backend": "nccl (RCCL)" if dist is not None else "none (world size 1)"},
        "roofline": {"bound": "hbm", "kernel": kname[0], "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms_avg": round(dec_ms, 4), "algorithmic_bytes_per_launch": alg},
        "rank_local_scan": {
            "what": "the same decode with every rank scanning its own slab and building its own index: "
                    "no collective, no serial whole-file scan on rank 0",
            "value": round(ncomplex * world * args.steps / elapsed_l / 1e6, 1),
            "unit": "M complex samples/s (threads x channels counted)",
            "ms_per_step": round(elapsed_l / args.steps * 1e3, 4),
            "index_equals_broadcast": bool(same_index.item() == 1.0)},
        "sanity_spot_check": ok}


def leg_api_read(args, image, out, kern_ms):
    """The headline image through the drop-in API: a stream reader opened on
    the device tensor, ``read(out=out)`` -- one scan / index / decode launch for
    the whole file (base/base.py:919-969 semantics; resident.py)."""
    from baseband_amd import vdif, _lib
    t_open = time.perf_counter()
    fh = vdif.open(image, 'rs', sample_rate=float(SPF * FRAME_RATE))
    open_ms = (time.perf_counter() - t_open) * 1e3
    assert fh.shape == (out.numel(),), (fh.shape, out.numel())
    fh.read(out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.steps):
        fh.seek(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fh.read(out=out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    kname = _lib.last_kernel()
    verify = fh.verify
    fh.close()
    ms = float(np.mean(ts))
    return {"call": "baseband_amd.vdif.open(<uint8 device tensor>, 'rs', sample_rate=32e6).read(out=<float32 device tensor>)",
            "ms": round(ms, 4), "ms_min": round(min(ts), 4), "open_ms": round(open_ms, 2),
            "value": round(out.numel() / ms / 1e3, 1), "unit": "Msamples/s",
            "verify": verify, "kernel": kname,
            "ms_over_kernel_leg": round(ms / kern_ms, 4),
            "timing": "host wall clock around read() incl. the verification sync, mean of {} calls".format(args.steps)}


def leg_invalid_fill(args, image, out, nframes, h0, first_frame, kern_ms):
    """SURVEY 8(d) "value distributions": the headline file with the
    `invalid_data` bit (word 0, bit 31) set in 1 % of the frames, through the
    same scan + index + decode step: flagged frames come out as the fill value
    (base/frame.py:191-199, vdif/frame.py:79-90) and the fill path costs
    nothing.  The bits are cleared again afterwards."""
    from baseband_amd import kernels, _lib
    pattern, mask = h0.invariant_pattern()
    g = torch.Generator(device=image.device)
    g.manual_seed(99)
    bad = torch.nonzero(torch.rand(nframes, generator=g, device=image.device) < 0.01).reshape(-1)
    w0 = image.view(torch.int32)[::FRAME_NBYTES // 4]
    w0[bad] |= -2 ** 31
    try:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        src = None
        for k in range(-1, args.steps):
            recs = kernels.vdif_scan(image, nframes, FRAME_NBYTES, HEADER_NBYTES, pattern, mask,
                                     h0['seconds'], h0['frame_nr'] + first_frame, FRAME_RATE)
            src = kernels.build_index(recs, nframes, 1, None)
            if k >= 0:
                ev[k][0].record()
            kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=src, out=out)
            if k >= 0:
                ev[k][1].record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        nbad = int(bad.numel())
        filled = int((src < 0).sum().item())
        # flagged frames are fill, their neighbours are data -- checked HERE, on
        # the output of the timed 1 %-invalid launches, before anything else
        # writes into `out` (base/frame.py:191-199)
        f = int(bad[nbad // 2].item())
        is_fill = bool((out[f * SPF:(f + 1) * SPF] == 0).all().item())
        g_ = f + 1 if f + 1 < nframes and not bool((bad == f + 1).any().item()) else max(0, f - 1)
        assert not bool((bad == g_).any().item()), "no unflagged neighbour to check"
        lev = _lib.get_levels(_lib.CODER_VDIF, 2)
        raw = image[g_ * FRAME_NBYTES + HEADER_NBYTES:(g_ + 1) * FRAME_NBYTES].cpu().numpy()
        neighbour_ok = bool(np.array_equal(out[g_ * SPF:(g_ + 1) * SPF].cpu().numpy().view(np.uint32),
                                           expand_2bit(raw, lev).view(np.uint32)))
        # a data frame is not all zeros (so `is_fill` above is not vacuous)
        neighbour_not_fill = not bool((out[g_ * SPF:(g_ + 1) * SPF] == 0).all().item())
        # every frame invalid: the same kernel with the same store pattern and NO
        # reads -- what this device does write-only ("measured achievable")
        none = torch.full_like(src, -1)
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        for a_, b_ in ev2:
            a_.record()
            kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src=none, out=out)
            b_.record()
        torch.cuda.synchronize()
        ms_w = float(np.median([a_.elapsed_time(b_) for a_, b_ in ev2][1:]))
        del none
        all_fill = bool((out[g_ * SPF:(g_ + 1) * SPF] == 0).all().item())
    finally:
        w0[bad] &= 2 ** 31 - 1
    alg = nframes * (FRAME_NBYTES + PAYLOAD_NBYTES * 16)
    return {"what": "the headline step with invalid_data set in 1 % of the frames (scan -> index entry -1 -> fill 0.0)",
            "frames_flagged": nbad, "index_entries_invalid": filled, "kernel": _lib.last_kernel(),
            "kernel_ms_avg": round(ms, 4), "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
            "ms_over_headline_kernel": round(ms / kern_ms, 4),
            "flagged_frame_is_fill": is_fill, "neighbour_frame_is_data": neighbour_ok and neighbour_not_fill,
            "checked": "on the output of the timed 1 %-invalid launches, before the all-invalid launch below",
            "all_frames_invalid": {"what": "the same launch with every index entry -1: the kernel's stores, no reads",
                                   "output_is_fill": all_fill,
                                   "kernel_ms": round(ms_w, 4),
                                   "write_GBps": round(nframes * PAYLOAD_NBYTES * 16 / ms_w / 1e6, 1),
                                   "frac_of_peak": round(nframes * PAYLOAD_NBYTES * 16 / ms_w / 1e6 / HBM_PEAK_GBS, 4),
                                   "headline_kernel_ms_over_this": round(kern_ms / ms_w, 4)}}


def leg_locate(image, h0, nframes, reps=5):
    """The corruption-tolerant frame search (base/base.py:181-335
    `locate_frames` as the `_bad_frame` recoveries use it) over the whole
    headline image: a read-only sweep that tests EVERY byte position against
    the header pattern and confirms hits one frame later (bb_vdif_locate,
    k_scan.h).  Algorithmic bytes = the file, read once."""
    import ctypes as C
    from baseband_amd import kernels, _lib
    pattern, mask = h0.invariant_pattern()
    p = kernels._vdif_params(FRAME_NBYTES, HEADER_NBYTES, pattern, mask, 0, 0, 0)
    nbytes = image.numel()
    cap = nbytes // FRAME_NBYTES + 16
    offs = torch.empty(cap, dtype=torch.int64, device=image.device)
    count = torch.zeros(1, dtype=torch.int64, device=image.device)

    def run():
        count.zero_()
        _lib.check(_lib.lib.bb_vdif_locate(image.data_ptr(), nbytes, C.byref(p), offs.data_ptr(), cap,
                                           count.data_ptr(), kernels._stream(image)), 'bb_vdif_locate')
    med, mean = timed_launches(run, reps)
    n = int(count.item())
    found = torch.sort(offs[:min(n, cap)]).values
    ok = n == nframes and bool((found == torch.arange(nframes, device=image.device, dtype=torch.int64)
                                * FRAME_NBYTES).all().item())
    return {"case": "bb_vdif_locate: byte-granular header search over the {:.3f} GiB cfg2 image".format(nbytes / 2 ** 30),
            "kernel": "k_vdif_locate (bb_locate_sweep)", "ms": round(mean, 4), "ms_median": round(med, 4),
            "timing": "incl. the memset of the hit counter",
            "algorithmic_GBps": round(nbytes / mean / 1e6, 1), "frac": round(nbytes / mean / 1e6 / HBM_PEAK_GBS, 4),
            "bytes_in": nbytes, "bytes_out": n * 8, "frames_found": n, "spot_check": ok,
            "spot_check_what": "every frame of the image found, at its offset, nothing else"}


def leg_other_configs(device, out, gib=8.0, gib8=31.0, reps=5, res=None):
    """Kernel-level figures for the other BASELINE configurations on random
    input -- `gib` GiB for the 2-bit formats, `gib8` for the 8-bit ones, i.e. the
    same 128-137 GB of decoded output as the headline launch each (the output
    of a launch should span as much of HBM as the headline's does: DESIGN.md 3.1,
    docs/DESIGN_rounds1-3.md "Where the output lies"): (ms, algorithmic GB/s, fraction of 8 TB/s,
    kernel as named by the library)."""
    from baseband_amd import kernels, _lib
    from baseband_amd.mark4._bitmaps import BITMAPS
    nbytes = int(gib * 2 ** 30)
    nbytes8 = int(min(gib8 * 2 ** 30, out.numel()))
    g = torch.Generator(device=device)
    g.manual_seed(4242)
    buf = torch.empty(max(nbytes, nbytes8) + 4096, dtype=torch.uint8, device=device)
    for lo in range(0, buf.numel() // 4, 1 << 28):
        hi = min(buf.numel() // 4, lo + (1 << 28))
        buf.view(torch.int32)[lo:hi] = torch.randint(-2 ** 31, 2 ** 31 - 1, (hi - lo,), generator=g,
                                                     device=device, dtype=torch.int64).to(torch.int32)
    res = [] if res is None else res            # (the caller's list keeps the rows measured before a failure)

    def add(name, fn, bytes_in, bytes_out, units, unit_name):
        med, mean = timed_launches(fn, reps)
        gbs = (bytes_in + bytes_out) / mean / 1e6
        res.append({"case": name, "kernel": _lib.last_kernel(), "ms": round(mean, 4),
                    "ms_median": round(med, 4), "algorithmic_GBps": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_in": bytes_in, "bytes_out": bytes_out,
                    "M{}_per_s".format(unit_name): round(units / mean / 1e3, 1)})

    # cfg0 layout: 8 threads x 1 channel 2-bit real, 5032-byte frames (sample.vdif)
    fn_, pn, nth = 5032, 5000, 8
    nsets = min(nbytes // (fn_ * nth), out.numel() // (nth * pn * 4))
    perm = torch.tensor([4, 0, 5, 1, 6, 2, 7, 3], device=device)
    pos = torch.arange(nsets, device=device, dtype=torch.int64)[:, None] * nth + perm[None, :]
    src = (pos * fn_ + 32).reshape(-1).contiguous()
    o = out[:nsets * nth * pn * 4]
    add("VDIF 8 threads x 1 channel 2-bit real (sample.vdif layout)",
        lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=1, nslot=nth, src=src, out=o),
        nsets * nth * fn_, o.numel() * 4, o.numel(), "samples")
    # cfg4a: Mark 5B 16 channels 2-bit
    nfr = min(nbytes // 10016, out.numel() // 40000)
    o = out[:nfr * 40000]
    add("Mark 5B 16 channels 2-bit",
        lambda: kernels.decode_frames(buf, nfr, 10000, _lib.CODER_MARK5B, 2, chunk=16, src0=16,
                                      src_stride=10016, out=o),
        nfr * 10016, o.numel() * 4, o.numel(), "samples")
    # cfg4b: Mark 4 64 tracks fanout 4
    m = BITMAPS[(8, 2, 4)]
    nfr = min(nbytes // 160000, out.numel() // (20000 * 32))
    o = out[:nfr * 20000 * 32]
    add("Mark 4 64 tracks fanout 4 (8 channels 2-bit)",
        lambda: kernels.decode_mark4(buf, nfr, 64, 20000, m['sign_bit'], m['mag_bit'], fill_words=160,
                                     src0=0, src_stride=160000, out=o),
        nfr * 160000, o.numel() * 4, o.numel(), "samples")
    # cfg5a: GUPPI 8-bit 2 pol complex 64 channels, channels first, 128 MiB blocks
    npol, nchan, blk = 2, 64, 128 << 20
    T = blk // (npol * nchan * 2)
    nbytes = nbytes8
    nfr = max(1, nbytes // blk)
    nb = nfr * T * npol * nchan * 2
    o = out[:nb]
    add("GUPPI 8-bit 2 pol 64 channels, channels first, OVERLAP 0",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T, src0=0,
                                        src_stride=blk, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    add("GUPPI 8-bit 2 pol 64 channels, time first",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npol, nchan, T, 0, T, src0=0,
                                        src_stride=blk, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    # cfg5b: DADA 8-bit 2 pol complex (flat int8) and MKBF heaps
    nb = nbytes // 4 * 4
    o = out[:nb]
    add("DADA 8-bit 2 pol complex (flat int8 -> float32)",
        lambda: kernels.decode_frames(buf, 1, nb, _lib.CODER_INT, 8, src0=0, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    nheap_t = 64
    Tm = 256 * nheap_t
    blkm = Tm * npol * nchan * 2
    nfr = max(1, nbytes // blkm)
    nb = nfr * blkm
    o = out[:nb]
    add("DADA MKBF heaps 2 pol 64 channels (256-sample heaps)",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_MKBF, npol, nchan, Tm, 0, Tm, src0=0,
                                        src_stride=blkm, out=o),
        nb, nb * 4, nb // 2, "complex_samples")
    # cfg5 "DADA float32 passthrough": NBIT 32 is an EXTENSION of this package
    # (the reference raises KeyError(32), dada/payload.py:40-41; parity is
    # unpinned by construction): the reader's `_decode_window` is ONE strided
    # copy launch of the library (bb_copy_frames, csrc/k_copy.h); here 128 MiB
    # payloads behind 4096-byte headers
    blk32 = 128 << 20
    nfr = max(1, min(nbytes8 - 4096, out.numel() * 4) // (blk32 + 4096))
    # the output where the reader puts it: a read() result of 1-64 GiB is a block of the
    # output arena (placement.empty_output); a slice of the headline tensor if that fails
    o, o_mem = None, "a slice of the 127.5 GiB headline tensor"
    try:
        import baseband_amd
        from baseband_amd import arena as _ar
        o = baseband_amd.empty_output((nfr * blk32 // 4,), dtype=torch.float32, device=device)
        a_ = _ar.default(device)
        o_mem = "arena block (placement.empty_output, as dada.open().read() allocates it)" \
            if a_ is not None and a_.owns(o) else "torch.empty"
    except Exception:
        o = None
    if o is None:
        o = out[:nfr * blk32 // 4]
    add("DADA NBIT=32 float32 passthrough (extension, parity unpinned: no reference counterpart)",
        lambda: kernels.copy_frames(buf, nfr, blk32, src0=4096, src_stride=blk32 + 4096, out=o),
        nfr * blk32, nfr * blk32, nfr * blk32 // 4, "samples")
    res[-1]["output_memory"] = o_mem
    k = nfr - 1
    res[-1]["spot_check"] = bool(torch.equal(
        o[k * (blk32 // 4):k * (blk32 // 4) + 4096].view(torch.int32),
        buf[4096 + k * (blk32 + 4096):4096 + k * (blk32 + 4096) + 16384].view(torch.int32)))
    del o

    # ---- the secondary kernels -------------------------
    def expand_bits(raw, lev, bps):
        """host re-expansion of packed codes, least significant field first"""
        sh = np.arange(0, 8, bps, dtype=np.uint8)
        return lev[(raw[:, None] >> sh) & ((1 << bps) - 1)].reshape(-1)

    def flat_case(name, coder, bps, frame, pay, hdr, limit):
        nfr_ = min(int(limit) // frame, out.numel() // (pay * 8 // bps))
        o_ = out[:nfr_ * (pay * 8 // bps)]
        add(name, lambda: kernels.decode_frames(buf, nfr_, pay, coder, bps, src0=hdr, src_stride=frame, out=o_),
            nfr_ * frame, o_.numel() * 4, o_.numel(), "samples")
        f = nfr_ - 1
        raw = buf[f * frame + hdr:f * frame + hdr + pay].cpu().numpy()
        got = o_[f * (pay * 8 // bps):(f + 1) * (pay * 8 // bps)].cpu().numpy()
        res[-1]["spot_check"] = bool(np.array_equal(
            got.view(np.uint32), expand_bits(raw, _lib.get_levels(coder, bps), bps).view(np.uint32)))

    flat_case("VDIF 1-bit real 1 channel, 8032-byte frames", _lib.CODER_VDIF, 1, 8032, 8000, 32, gib * 2 ** 30)
    flat_case("VDIF 4-bit real 1 channel, 8032-byte frames", _lib.CODER_VDIF, 4, 8032, 8000, 32, 2 * gib * 2 ** 30)
    flat_case("VDIF 8-bit real 1 channel, 8032-byte frames", _lib.CODER_VDIF, 8, 8032, 8000, 32, nbytes8)
    flat_case("GSB rawdump 4-bit real (2^22-byte blocks, no headers)", _lib.CODER_INT, 4, 1 << 22, 1 << 22, 0,
              2 * gib * 2 ** 30)
    # a reader `subset` of 2 of 16 channels folded into the decode of 8-thread
    # 16-channel complex VDIF (k_decode_gather_select); bytes moved = every frame
    # read + the kept channels written
    nth, nch, pn, fn_ = 8, 16, 8000, 8032
    nsets = int(gib * 2 ** 30) // (fn_ * nth)
    src = (torch.arange(nsets * nth, device=device, dtype=torch.int64) * fn_ + 32).contiguous()
    within = torch.tensor([6, 7, 24, 25], dtype=torch.int32, device=device)          # channels 3 and 12 (re, im)
    spf = pn * 4 // (2 * nch)
    o = out[:nsets * spf * nth * 4]
    add("VDIF 8 threads x 16 channels 2-bit complex, subset of 2 of 16 channels folded into the decode",
        lambda: kernels.decode_frames(buf, nsets, pn, _lib.CODER_VDIF, 2, chunk=2 * nch, nslot=nth, src=src,
                                      complex_data=True, out=o, within=within),
        nsets * nth * fn_, o.numel() * 4, o.numel() // 2, "complex_samples")
    full = kernels.decode_frames(buf, 1, pn, _lib.CODER_VDIF, 2, chunk=2 * nch, nslot=nth,
                                 src=src[(nsets - 1) * nth:], complex_data=True)
    res[-1]["spot_check"] = bool(torch.equal(
        full.view(spf, nth, 2 * nch)[:, :, within.long()].reshape(-1).view(torch.int32),
        o[(nsets - 1) * spf * nth * 4:].view(torch.int32)))
    del full
    # a channel LIST (8 scattered of 64) of time-first GUPPI blocks
    npol, nchan, blk = 2, 64, 128 << 20
    T = blk // (npol * nchan * 2)
    nfr = max(1, int(gib * 2 ** 30) // blk)
    cmap = torch.tensor([1, 5, 9, 20, 33, 40, 41, 63], dtype=torch.int32, device=device)
    nsel = int(cmap.numel())
    o = out[:nfr * T * npol * nsel * 2]
    add("GUPPI 8-bit 2 pol 64 channels, time first, channel list of 8 of 64 (bytes moved: every block read, kept channels written)",
        lambda: kernels.decode_i8_tiled(buf, nfr, _lib.LAYOUT_GUPPI_TF, npol, nsel, T, 0, T, src0=0, src_stride=blk,
                                        out=o, nchan_stored=nchan, npol_stored=npol, chan_map=cmap),
        nfr * blk, o.numel() * 4, o.numel() // 2, "complex_samples")
    full = kernels.decode_i8_tiled(buf, 1, _lib.LAYOUT_GUPPI_TF, npol, nchan, 4096, 0, 4096, src0=(nfr - 1) * blk,
                                   src_stride=blk)
    res[-1]["spot_check"] = bool(torch.equal(
        full.view(4096, npol, nchan, 2)[:, :, cmap.long()].reshape(-1).view(torch.int32),
        o[(nfr - 1) * T * npol * nsel * 2:][:4096 * npol * nsel * 2].view(torch.int32)))
    del full
    # the encoders: float32 -> packed codes; 4 B read per sample
    from baseband_amd._lib import lib as _L, check as _check
    for bps_, coder_ in ((2, _lib.CODER_VDIF), (4, _lib.CODER_VDIF), (8, _lib.CODER_VDIF)):
        nval = min(out.numel(), (buf.numel() - 4096) * 8 // bps_) // 1024 * 1024
        vals = out[:nval]
        packed = buf[:nval * bps_ // 8]

        def enc():
            _check(_L.bb_encode_flat(vals.data_ptr(), nval, coder_, bps_, packed.data_ptr(), packed.numel(),
                                     kernels._stream(vals)), 'bb_encode_flat')
        if bps_ == 2:
            # `out` holds decoded 2-bit levels nowhere in particular by now: fill its
            # head with a decode, so that encode(decode(x)) == x can be checked
            nchk = 4096
            raw0 = buf[:nchk * 8032].clone()
            kernels.decode_frames(raw0, nchk, 8000, _lib.CODER_VDIF, 2, src0=32, src_stride=8032, out=out[:nchk * 32000])
        add("bb_encode_flat VDIF {}-bit (float32 -> packed codes)".format(bps_), enc, nval * 4, nval * bps_ // 8,
            nval, "samples")
        if bps_ == 2:
            res[-1]["spot_check"] = bool(torch.equal(
                packed[:nchk * 8000].view(nchk, 8000), raw0.view(nchk, 8032)[:, 32:]))
            res[-1]["spot_check_what"] = "encode(decode(x)) == x on 4096 payloads"
            del raw0
    return res


def pinned_h2d_rate(device, nbytes=1 << 30, reps=5):
    """The link: one pinned buffer -> HBM with hipMemcpyAsync, GB/s (median)."""
    host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dev.copy_(host, non_blocking=True)
        b.record()
        b.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    del host, dev
    return nbytes / float(np.median(ts)) / 1e6


def leg_pipeline(device, gib=2.0, reads=3):
    """The PCIe-inclusive path (north_star: "overlapped with pinned
    hipMemcpyAsync of the next file chunk on a side stream"; SURVEY 8(d) cfg5;
    replaces the per-frame ``fh.read`` of base/payload.py:122-137): files of
    `gib` GiB written with this package's own stream writers, page cache warm,
    then ``open(path).read()`` with the defaults a user gets (verify on) --
    windows of whole frame sets go page cache -> pinned buffer -> HBM on a side
    stream while the window before them is scanned and decoded.  Reported per
    format: GB/s of FILE bytes (best and median of `reads` reads incl. open and
    close), the ratio to the pinned H2D rate measured here, the per-window
    times of one traced read (host copy, host wait for a buffer, H2D and
    kernels by events), and a check: the windowed read equals, bit for bit, the
    decode of the same file bytes resident in HBM (one scan / decode launch)."""
    import baseband_amd as bb
    from baseband_amd import staging
    nbytes = int(gib * 2 ** 30)
    tmp_root = os.environ.get('TMPDIR', '/tmp')
    try:
        free_b = shutil.disk_usage(tmp_root).free
    except OSError:
        free_b = 0
    if free_b < nbytes + (1 << 30):
        # (an environment matter, not a result: the leg is skipped and counts for no check)
        return {"skipped": "{} has {:.1f} GiB free, a {:.1f} GiB temporary file does not fit".format(
            tmp_root, free_b / 2 ** 30, gib)}
    tmp = tempfile.mkdtemp(prefix='bb_pipe_', dir=tmp_root)
    g = torch.Generator(device=device)
    g.manual_seed(2718)
    t0 = np.datetime64('2014-06-13T05:30:01')
    link = pinned_h2d_rate(device)
    res = {"file_GiB_each": gib, "pinned_h2d_GBps": round(link, 2),
           "what": "open(path).read() of a file in the page cache, defaults (verify on); GB/s of file bytes",
           "formats": []}

    def write(opener, chunk, nchunks):
        with opener() as fw:
            for _ in range(nchunks):
                fw.write(chunk)

    def case(name, path, writer, reader_kw, opener):
        row = {"case": name}
        try:
            tw = time.perf_counter()
            writer()
            row["write_s"] = round(time.perf_counter() - tw, 3)
            size = os.path.getsize(path)
            # (the stream writer: GPU encode -> pinned -> one write() per 16 MiB; a buffered
            # write() into a new file is what bounds it, 11-12 GB/s on this host class:
            # profiles/r03y_exp_file_write.log)
            row["writer_GBps"] = round(size / max(row["write_s"], 1e-9) / 1e9, 2)
            with open(path, 'rb') as f:                      # warm the page cache
                while f.read(64 << 20):
                    pass
            ts, parts = [], []
            got = None
            for r in range(reads + 1):
                del got
                torch.cuda.synchronize()
                t = time.perf_counter()
                fh = opener(path, 'rs', **reader_kw)
                t_open = time.perf_counter()
                got = fh.read()
                t_read = time.perf_counter()
                torch.cuda.synchronize()
                t_sync = time.perf_counter()
                fh.close()
                t_end = time.perf_counter()
                if r:
                    ts.append(t_end - t)
                    parts.append((t_open - t, t_read - t_open, t_sync - t_read, t_end - t_sync))
            # one more, traced per window
            del got
            staging.trace = []
            try:
                t = time.perf_counter()
                with opener(path, 'rs', **reader_kw) as fh:
                    got = fh.read()
                torch.cuda.synchronize()
                traced_s = time.perf_counter() - t
                summary = staging.window_trace_summary(staging.trace)
            finally:
                staging.trace = None
            # the same bytes resident in HBM: one scan / index / decode launch
            with open(path, 'rb') as f:
                raw = np.frombuffer(f.read(), np.uint8)
            dev = torch.from_numpy(raw.copy()).to(device)
            with opener(dev, 'rs', **reader_kw) as fh:
                ref = fh.read()
            same = bool(got.shape == ref.shape and torch.equal(
                torch.view_as_real(got).view(torch.int32) if got.is_complex() else got.view(torch.int32),
                torch.view_as_real(ref).view(torch.int32) if ref.is_complex() else ref.view(torch.int32)))
            best, med = min(ts), float(np.median(ts))
            row.update({"file_bytes": size, "shape": list(got.shape), "read_s_best": round(best, 4),
                        "file_GBps_best": round(size / best / 1e9, 2), "file_GBps_median": round(size / med / 1e9, 2),
                        "fraction_of_pinned_h2d": round(size / best / 1e9 / link, 3),
                        "host_ms_of_the_best_read": dict(zip(("open", "read_call", "final_sync", "close"),
                                                             [round(x * 1e3, 2) for x in parts[int(np.argmin(ts))]])),
                        "host_ms_of_each_read": [[round(x * 1e3, 2) for x in p_] for p_ in parts],
                        "windows": summary, "traced_read_s": round(traced_s, 4),
                        "equals_resident_decode": same})
            del got, ref, dev, raw
        except Exception as exc:
            row["error"] = repr(exc)[:400]
        finally:
            try:
                os.remove(path)
            except OSError:
                pass
        res["formats"].append(row)

    try:
        from baseband_amd.vdif.header import VDIFHeader
        # cfg2: VDIF 1 thread 2-bit real
        path = os.path.join(tmp, 'cfg2.vdif')
        nfr = nbytes // 8032
        per = 4096
        chunk = torch.randn(per * 32000, device=device, generator=g) * 2.
        h0 = VDIFHeader.fromvalues(edv=0, time=t0, nchan=1, bps=2, complex_data=False, thread_id=0,
                                   samples_per_frame=32000, station='AA')
        case("VDIF cfg2 (1 thread, 2-bit real, 8032-byte frames)", path,
             lambda: write(lambda: bb.vdif.open(path, 'ws', header0=h0, sample_rate=32e6, nthread=1), chunk, nfr // per),
             dict(sample_rate=32e6), bb.vdif.open)
        del chunk
        # cfg3: VDIF 8 threads x 16 channels 2-bit complex
        path = os.path.join(tmp, 'cfg3.vdif')
        nsets = nbytes // (8032 * 8)
        per = 1024
        chunk = torch.view_as_complex(torch.randn(per * 1000, 8, 16, 2, device=device, generator=g) * 2.)
        h3 = VDIFHeader.fromvalues(edv=0, time=t0, nchan=16, bps=2, complex_data=True, thread_id=0,
                                   samples_per_frame=1000, station='AA')
        case("VDIF cfg3 (8 threads x 16 channels, 2-bit complex)", path,
             lambda: write(lambda: bb.vdif.open(path, 'ws', header0=h3, sample_rate=1e6, nthread=8), chunk, nsets // per),
             dict(sample_rate=1e6), bb.vdif.open)
        del chunk
        # Mark 5B 16 channels 2-bit
        path = os.path.join(tmp, 'x.m5b')
        nfr = nbytes // 10016
        per = 4096
        chunk = torch.randn(per * 2500, 16, device=device, generator=g) * 2.
        case("Mark 5B 16 channels 2-bit", path,
             lambda: write(lambda: bb.mark5b.open(path, 'ws', sample_rate=32e6, nchan=16, bps=2, time=t0), chunk, nfr // per),
             dict(sample_rate=32e6, nchan=16, kday=56000), bb.mark5b.open)
        del chunk
        # Mark 4 64 tracks fanout 4
        path = os.path.join(tmp, 'x.m4')
        nfr = nbytes // 160000
        per = 256
        chunk = torch.randn(per * 80000, 8, device=device, generator=g) * 2.
        case("Mark 4 64 tracks fanout 4", path,
             lambda: write(lambda: bb.mark4.open(path, 'ws', sample_rate=32e6, ntrack=64, bps=2, fanout=4, time=t0),
                           chunk, nfr // per),
             dict(ntrack=64, decade=2010, sample_rate=32e6), bb.mark4.open)
        del chunk
        # GUPPI 8-bit 2 pol 64 channels, 128 MiB blocks
        from baseband_amd.guppi.header import GUPPIHeader
        path = os.path.join(tmp, 'x.raw')
        spf = (128 << 20) // (2 * 64 * 2)
        hg = GUPPIHeader.fromvalues(time=t0, sample_rate=1e6, samples_per_frame=spf, overlap=0,
                                    npol=2, nchan=64, pktsize=8192, bps=8)
        chunk = torch.view_as_complex(torch.randn(spf, 2, 64, 2, device=device, generator=g) * 30.)
        case("GUPPI 8-bit 2 pol 64 channels, 128 MiB blocks", path,
             lambda: write(lambda: bb.guppi.open(path, 'ws', header0=hg), chunk, nbytes // (128 << 20)),
             dict(), bb.guppi.open)
        del chunk
        # DADA 8-bit 2 pol complex, 128 MiB frames
        from baseband_amd.dada.header import DADAHeader
        path = os.path.join(tmp, 'x.dada')
        spf = (128 << 20) // 4
        hd = DADAHeader.fromvalues(time=t0, sample_rate=16e6, bps=8, complex_data=True, npol=2, nchan=1,
                                   samples_per_frame=spf)
        chunk = torch.view_as_complex(torch.randn(spf, 2, 2, device=device, generator=g) * 30.)
        case("DADA 8-bit 2 pol complex, 128 MiB frames", path,
             lambda: write(lambda: bb.dada.open(path, 'ws', header0=hd), chunk, nbytes // (128 << 20)),
             dict(), bb.dada.open)
        del chunk
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        staging.release_pinned()
    res["all_match"] = bool(res["formats"]) and all(f.get("equals_resident_decode") is True for f in res["formats"])
    return res


def leg_mid_size(device, image, draws=5, launches=6):
    """the launch sizes an ordinary ``read()`` issues
    (/root/reference semantics: base/base.py:919-969) -- cfg2 windows of 2^15,
    2^16 and 2^18 frames (4.2, 8.4, 33.6 GB of output) and a GUPPI
    channels-first read of 8 GiB (34 GB of output).  Every draw is a FRESH
    output: `draws` new ``torch.empty`` allocations (the cache emptied in
    between, so each is a new piece of HBM), and `draws` new blocks from the
    placement arena -- what the readers allocate from by default
    (baseband_amd/placement.py) -- taking turns.  Every launch decodes the NEXT
    window of the 8 GiB image (nothing of the input can still be in the 256
    MiB Infinity Cache); a draw's figure is the median of `launches` launches by
    HIP events on the launching stream; reported: min / median / max over the
    draws of the fraction of 8 TB/s.  Also one timed pass through the drop-in
    API per size: ``fh.read(count)`` allocating its own output."""
    import baseband_amd
    from baseband_amd import kernels, _lib, arena, placement, vdif
    img_frames = image.numel() // FRAME_NBYTES
    nxt = [0]
    peak_use = [0]

    def rate(out, nf):
        if ar is not None:
            peak_use[0] = max(peak_use[0], int(ar.stats()["bytes_in_use"]))
        ts = []
        for r in range(launches + 1):
            if nxt[0] + nf > img_frames:
                nxt[0] = 0
            first = nxt[0]
            nxt[0] += nf
            win = image[first * FRAME_NBYTES:(first + nf) * FRAME_NBYTES]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            kernels.decode_frames(win, nf, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2, src0=HEADER_NBYTES,
                                  src_stride=FRAME_NBYTES, out=out)
            b.record()
            b.synchronize()
            if r:
                ts.append(a.elapsed_time(b))
        return nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / float(np.median(ts)) / 1e6      # GB/s

    def summary(v):
        f = np.array(v) / HBM_PEAK_GBS
        return {"GBps_per_draw": [round(x, 1) for x in v],
                "frac_min": round(float(f.min()), 4), "frac_median": round(float(np.median(f)), 4),
                "frac_max": round(float(f.max()), 4)}

    torch.cuda.empty_cache()
    # the arena the readers create on their first large output (placement.py);
    # registered as an open reader for the length of this leg: an arena whose last
    # block dies while no reader is open gives its memory back, and every draw
    # below would grow (and probe) a new step
    placement.reader_opened()
    ar = placement._arena_for(device)
    res = {"arena": None if ar is None else ar.stats(),
           "method": "fresh output per draw ({} draws, torch.empty and arena blocks taking turns); per draw the median of "
                     "{} launches, each on the next window of the resident 8 GiB image; HIP events".format(draws, launches),
           "sizes": []}
    for lf in (15, 16, 18):
        nf = 1 << lf
        n = nf * SPF
        v_t, v_a, held = [], [], []
        for d in range(draws):
            o = torch.empty(n, dtype=torch.float32, device=device)
            v_t.append(rate(o, nf))
            del o
            torch.cuda.empty_cache()
            o = ar.empty(n) if ar is not None else None
            if o is None and held:
                held.clear()                         # the arena is full of the pieces held back: start over
                o = ar.empty(n)
            if o is not None:
                v_a.append(rate(o, nf))
                held.append(ar.empty((64 << 20) // 4))      # so that the next block starts elsewhere
                del o
        del held
        row = {"frames": nf, "output_GB": round(n * 4 / 1e9, 2), "kernel": _lib.last_kernel(),
               "torch_empty": summary(v_t), "arena": summary(v_a) if v_a else None}
        # the drop-in API on the same image: read(count) allocates its own output
        try:
            with vdif.open(image, 'rs', sample_rate=float(SPF * FRAME_RATE)) as fh:
                ts = []
                for k in range(4):
                    fh.seek(((k * 3 + 1) * nf % (img_frames - nf)) * SPF)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    got = fh.read(nf * SPF)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                    inside = ar is not None and ar.owns(got)
                    del got
                ms = float(np.median(ts[1:])) * 1e3
                # the same calls back to back, no host sync in between: read()
                # returns once its frames are verified, the decode goes on behind it
                nb2b = 8
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(nb2b):
                    fh.seek(((k * 5 + 2) * nf % (img_frames - nf)) * SPF)
                    got = fh.read(nf * SPF)
                    del got
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                ms_b2b = (time.perf_counter() - t0) / nb2b * 1e3
                row["api_read"] = {"call": "fh.read({} * 32000) at changing offsets, output allocated by the reader".format(nf),
                                   "ms_median": round(ms, 3), "output_in_arena": bool(inside),
                                   "GBps": round(nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / ms / 1e6, 1),
                                   "frac": round(nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / ms / 1e6 / HBM_PEAK_GBS, 4),
                                   "timing": "host wall clock incl. scan, index, allocation and the verification sync",
                                   "back_to_back": {"reads": nb2b, "ms_per_read": round(ms_b2b, 3),
                                                    "host_ms_per_read": round(t_host / nb2b * 1e3, 3),
                                                    "frac": round(nf * (FRAME_NBYTES + PAYLOAD_NBYTES * 16) / ms_b2b / 1e6 / HBM_PEAK_GBS, 4),
                                                    "what": "the same read() calls without a host sync in between"}}
        except Exception as exc:
            row["api_read"] = {"error": repr(exc)[:300]}
        res["sizes"].append(row)
    # GUPPI channels first, 8 GiB in -> 34 GB out (the first 8 GiB of the image as 64 blocks of 128 MiB)
    try:
        npol, nchan, blk = 2, 64, 128 << 20
        T = blk // (npol * nchan * 2)
        nfr = min(64, image.numel() // blk)
        nb = nfr * blk
        v_t, v_a = [], []

        def grate(o):
            med, mean = timed_launches(lambda: kernels.decode_i8_tiled(image, nfr, _lib.LAYOUT_GUPPI_CF, npol, nchan, T, 0, T,
                                                                       src0=0, src_stride=blk, out=o), launches)
            return (nb + nb * 4) / med / 1e6
        for d in range(draws):
            o = torch.empty(nb, dtype=torch.float32, device=device)
            v_t.append(grate(o))
            del o
            torch.cuda.empty_cache()
            o = ar.empty(nb) if ar is not None else None
            if o is not None:
                v_a.append(grate(o))
                del o
        res["guppi_cf_8GiB_in"] = {"output_GB": round(nb * 4 / 1e9, 2), "kernel": _lib.last_kernel(),
                                   "torch_empty": summary(v_t), "arena": summary(v_a) if v_a else None}
    except Exception as exc:
        res["guppi_cf_8GiB_in"] = {"error": repr(exc)[:300]}
    if ar is not None:
        st = ar.stats()
        res["arena_after"] = st
        res["arena_bytes_backed_per_byte_in_use_peak"] = (
            round(st["bytes_backed"] / max(1, peak_use[0]), 2) if peak_use[0] else None)
        res["arena_peak_bytes_in_use"] = peak_use[0]
    placement.reader_closed()           # (the arena trims itself now: nothing of it is alive)
    if ar is not None:
        res["arena_bytes_backed_after_last_reader_closed"] = ar.stats()["bytes_backed"]
    return res


# ------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--gib', type=float, default=8.0, help="file image size per GPU")
    ap.add_argument('--cfg3-gib', type=float, default=8.0, help="cfg3 leg: file bytes per GPU")
    ap.add_argument('--pipeline-gib', type=float, default=2.0, help="pipeline leg: size of each temporary file")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--traffic', choices=('live', 'file', 'none'), default='live')
    ap.add_argument('--no-extra-legs', action='store_true',
                    help="skip api_read / cfg3 / other_configs (headline only); same as --legs headline")
    ap.add_argument('--legs', choices=('all', 'headline', 'cfg3'), default='all',
                    help="all (default); headline: the headline step only, no CPU baseline, no counter passes "
                         "(an N-GPU run finishes in well under a minute); cfg3: headline + the cfg3 leg")
    ap.add_argument('--pmc-child', action='store_true',
                    help="internal: headline kernel only, no JSON extras (run under rocprofv3 --pmc)")
    ap.add_argument('--force-dist', action='store_true',
                    help="initialise the RCCL process group even with one rank, so that every "
                         "collective of the multi-rank path runs (a check of the N > 1 code on a 1-GPU box)")
    ap.add_argument('--dry-run', action='store_true',
                    help="CPU rehearsal of the multi-rank plumbing with gloo (no GPU, no measurement)")
    args = ap.parse_args()
    if args.pmc_child:
        args.no_cpu_baseline, args.no_extra_legs, args.traffic = True, True, 'none'
    if args.legs == 'headline':
        args.no_cpu_baseline, args.no_extra_legs, args.traffic = True, True, 'none'
    elif args.legs == 'cfg3':
        args.no_cpu_baseline, args.traffic = True, 'none'

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # not under a launcher: become one (nothing here has touched the GPU)
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.dry_run:
        return dry_run(args, rank, world)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus {} but WORLD_SIZE is {}".format(args.gpus, world))
    # host-side legs run first: they fork / start child processes, which must
    # happen before this process initialises the GPU
    cpu = None
    traffic_detail = None
    # (is there a GPU?  Asked WITHOUT initialising the HIP runtime: the legs
    # below fork workers and start profiler children, and a forked child must
    # not inherit a live runtime -- torch.cuda.device_count() can initialise it
    # on this build)
    have_gpu = os.path.exists('/dev/kfd')
    if rank == 0 and have_gpu and not args.no_cpu_baseline:
        handed = os.environ.get(CPU_JSON_ENV)
        if handed and os.path.exists(handed):
            with open(handed) as f:
                cpu = json.load(f)                  # timed by the parent that started the ranks
        else:
            # N = 1, or N > 1 under an external launcher: rank 0 times it before it
            # touches the GPU (the other ranks wait in the rendezvous meanwhile)
            cpu = cpu_baseline()
            if world > 1:
                cpu["timed_by"] = "rank 0 before initialising the GPU; the other ranks waited in the rendezvous"
    if rank == 0 and world > 1:
        traffic_detail = {"hbm_bytes_per_launch": None,
                          "reason": "counter passes run at N = 1 only: they are child runs of this script "
                                    "under rocprofv3 --pmc on the first GPU, which the ranks of an N > 1 "
                                    "run are about to use; the N = 1 line of the same commit carries the counters"}
    if rank == 0 and world == 1 and have_gpu:
        if abs(args.gib - 8.0) < 1e-9 and args.traffic != 'none':
            if args.traffic == 'live':
                try:
                    traffic_detail = live_traffic(args.gib)
                except Exception as exc:
                    traffic_detail = {"live_error": repr(exc)[:400]}
            if traffic_detail is None or "hbm_bytes_per_launch" not in traffic_detail:
                try:
                    err = (traffic_detail or {}).get("live_error")
                    traffic_detail = file_traffic()
                    if err:
                        traffic_detail["live_error"] = err
                except Exception:
                    pass
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback).")
    if world > torch.cuda.device_count():
        raise SystemExit("bench.py: {} ranks but only {} GPUs visible".format(world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        import datetime
        # (a rank that fails must become an error on the others, not a hang)
        # (RCCL prints a version banner on the C library's stdout when the communicator is
        # made; the contract is ONE JSON line there: the banner goes to stderr)
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', device_id=device, timeout=datetime.timedelta(seconds=240))
            warm = torch.zeros(1, device=device)
            dist.all_reduce(warm)               # (communicators are made lazily on some builds)
            torch.cuda.synchronize()
        finally:
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(saved)

    from baseband_amd import kernels, _lib
    from baseband_amd.parallel import frame_slab
    kernels.init()

    nframes = int(args.gib * 2 ** 30) // FRAME_NBYTES
    # time-slab sharding: rank r owns frames [r*nframes, (r+1)*nframes)
    first_frame, _ = frame_slab(nframes * world, rank, world)
    image, image_memory = image_buffer(nframes * FRAME_NBYTES, device)
    image, h0 = make_file_image_on_device(nframes, 12345 + rank, first_frame, device, into=image)
    pattern, mask = h0.invariant_pattern()
    out = empty_with_patience(nframes * SPF, torch.float32, device)
    bytes_in = nframes * FRAME_NBYTES
    bytes_out = nframes * SPF * 4
    alg_bytes = bytes_in + bytes_out

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(k=None):
        recs = kernels.vdif_scan(image, nframes, FRAME_NBYTES, HEADER_NBYTES, pattern,
                                 mask, h0['seconds'], h0['frame_nr'] + first_frame,
                                 FRAME_RATE)
        src = kernels.build_index(recs, nframes, 1, None)
        if k is not None:
            ev[k][0].record()
        kernels.decode_frames(image, nframes, PAYLOAD_NBYTES, _lib.CODER_VDIF, 2,
                              src=src, out=out)
        if k is not None:
            ev[k][1].record()
        return recs, src

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    kernel_name = _lib.last_kernel()

    # sanity spot check (outside the timed region; NOT the parity proof, which
    # lives in tests/): three frames re-expanded on the host from the library's
    # own level table
    lev = _lib.get_levels(_lib.CODER_VDIF, 2)
    ok = True
    if args.warmup:
        for f in (0, nframes // 3, nframes - 1):
            raw = image[f * FRAME_NBYTES + HEADER_NBYTES:(f + 1) * FRAME_NBYTES].cpu().numpy()
            got = out[f * SPF:(f + 1) * SPF].cpu().numpy()
            ok &= bool(np.array_equal(got.view(np.uint32), expand_2bit(raw, lev).view(np.uint32)))

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg = sum(kern_ms) / len(kern_ms)
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        allk = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(allk, torch.tensor([kern_avg], dtype=torch.float64, device=device))
        per_rank = [float(x.item()) for x in allk]
    if args.pmc_child:
        return

    achieved = alg_bytes / (kern_avg * 1e-3) / 1e9
    total_samples = nframes * SPF * world * args.steps
    value = total_samples / elapsed / 1e6
    traffic = None
    if traffic_detail and traffic_detail.get("hbm_bytes_per_launch"):
        traffic = traffic_detail["hbm_bytes_per_launch"]
        traffic_detail["traffic_over_algorithmic"] = round(traffic / alg_bytes, 4)
        if traffic < 0.98 * alg_bytes:
            # fewer bytes than the kernel must move: the counter rows are not the
            # headline launches' (say so rather than report an impossible figure)
            traffic_detail["rejected"] = "counters below the algorithmic bytes: {:.4g} B".format(traffic)
            traffic = None

    line = {
        "metric": "decoded Msamples/s, VDIF 2-bit 1-thread (scan + index + decode, input resident in HBM)",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "float32", "data": "synthetic",
        "config": {"workload": "cfg2: synthetic {:.3f} GiB per GPU single-thread VDIF, "
                               "2-bit real, 1 channel, EDV 0, 8032-byte frames"
                               .format(bytes_in / 2 ** 30),
                   "frames_per_gpu": nframes, "bytes_in_per_gpu": bytes_in,
                   "bytes_out_per_gpu": bytes_out,
                   "input": "packed 2-bit codes (uint8 file image; memory: {})".format(image_memory),
                   "output": "float32 samples: "
                   "full-size tensor kept in HBM (no slab recycling)",
                   "sharding": "time slabs, one per rank, no collective"},
        "roofline": {"bound": "hbm", "kernel": kernel_name,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms_avg": round(kern_avg, 4),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "traffic": traffic, "traffic_detail": traffic_detail},
        "sanity_spot_check": ok,
    }
    if per_rank is not None:
        line["per_rank"] = {"kernel_ms_avg": [round(x, 4) for x in per_rank],
                            "roofline_frac": [round(alg_bytes / (x * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                              for x in per_rank]}
    if not args.no_extra_legs:
        # (each leg is fenced: a failure is reported in its slot, the headline stands)
        if rank == 0 and world > 1:
            # the legs below include a collective; should a rank die inside it
            # the watchdog takes this process down before the line is printed,
            # so leave the measured headline in the log first (stderr: the
            # contract's ONE line on stdout is printed at the end)
            print("bench.py: headline before the extra legs: " + json.dumps(line),
                  file=sys.stderr, flush=True)
        if rank == 0 and args.legs == 'all':
            try:
                line["parity_digests"] = parity_digests()
            except Exception as exc:
                line["parity_digests"] = {"error": repr(exc)[:300]}
            try:
                line["api_read"] = leg_api_read(args, image, out, kern_avg)
            except Exception as exc:
                line["api_read"] = {"error": repr(exc)[:500]}
            try:
                line["invalid_fill"] = leg_invalid_fill(args, image, out, nframes, h0, first_frame, kern_avg)
                # the fraction of the spec peak AND of what the device
                # was measured to do -- this launch's stores without its reads
                w = line["invalid_fill"]["all_frames_invalid"]
                line["roofline"]["measured_write_only"] = {
                    "GBps": w["write_GBps"], "frac_of_peak": w["frac_of_peak"],
                    "kernel_time_over_write_only_time": w["headline_kernel_ms_over_this"],
                    "what": "the headline launch with every index entry -1 (fill): same kernel, same stores, no reads"}
            except Exception as exc:
                line["invalid_fill"] = {"error": repr(exc)[:500]}
        locate = None
        if rank == 0 and world == 1 and args.legs == 'all':
            try:
                locate = leg_locate(image, h0, nframes)
            except Exception as exc:
                locate = {"case": "bb_vdif_locate", "error": repr(exc)[:500]}
        del image
        try:
            line["cfg3"] = leg_cfg3(args, rank, world, device, dist, out)
        except Exception as exc:
            # (with several ranks the others run into the collective's timeout
            # and land here too; the headline measured above still gets printed)
            line["cfg3"] = {"error": repr(exc)[:500]}
        if rank == 0 and world == 1 and args.legs == 'all':
            try:
                line["other_configs"] = []
                leg_other_configs(device, out, res=line["other_configs"])
            except Exception as exc:
                line["other_configs"].append({"error": repr(exc)[:500]})
            if locate is not None:
                line["other_configs"].append(locate)
            # the launch sizes of ordinary read() calls, into fresh outputs: the
            # 127.5 GiB headline output goes first, the 8 GiB image comes back
            del out
            torch.cuda.empty_cache()
            try:
                image, _ = image_buffer(nframes * FRAME_NBYTES, device)
                image, _ = make_file_image_on_device(nframes, 12345, 0, device, into=image)
                line["mid_size"] = leg_mid_size(device, image)
                del image
            except Exception as exc:
                line["mid_size"] = {"error": repr(exc)[:500]}
            torch.cuda.empty_cache()
            try:
                line["pipeline"] = leg_pipeline(device, gib=args.pipeline_gib)
            except Exception as exc:
                line["pipeline"] = {"error": repr(exc)[:500]}
    rc = 0
    if rank == 0:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        rc = finish(line)
    if dist is not None:
        dist.destroy_process_group()
    return rc


if __name__ == '__main__':
    sys.exit(main() or 0)
