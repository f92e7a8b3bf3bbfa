"""cpu_baseline: the NumPy port of the reference loop timed on the host cores."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import *          # noqa: F401,F403
from .common import _s32, _git_commit, _run_group, _free_port     # noqa: F401

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
              "sample_short": "{} x {} cfg2 frames, oracle/bb_oracle_np.vdif_read".format(reps, nframes),
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

