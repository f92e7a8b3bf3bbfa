#!/opt/conda/bin/python3.9
"""Calibrate bench.py's CPU baseline against the real reference.

bench.py times ``oracle/bb_oracle_np.vdif_read`` (a NumPy restatement, kind
"port") because the reference cannot travel to the GPU box.  The port leaves
out the reference's per-frame object construction (VDIFHeader / VDIFPayload /
VDIFFrameSet per frame set: base/base.py:957-967, vdif/frame.py:176-243), so it
is faster than the reference as written.  This script measures how much, HERE
(development container, /root/reference importable): the same seeded cfg2 file
(BASELINE.json configs[1] layout: single-thread VDIF, 2-bit real, 1 channel,
EDV 0, 8032-byte frames) is decoded by

  * the real ``baseband.vdif.open(name, 'rs').read()`` with verify=True and
    verify=False,
  * the port,

pinned to ONE core, REPEATS rounds in which the three take turns (so that
whatever else the host is doing hits all of them alike); median and
inter-quartile range are reported and the ratios are ratios of MEDIANS
(the fastest-of-11 figures of round 2 rested on two minima of a
0.2-4.4 s spread).  Outputs are compared bit for bit.  Result:
tests/golden/cpu_calibration.json (SURVEY.md
section 8d).  bench.py copies the ratio into ``cpu_baseline.calibration``.

    /opt/conda/bin/python3.9 -W ignore tools/calibrate_cpu_baseline.py
"""
import json
import os
import platform
import sys
import tempfile
import time

import numpy as np

np.asscalar = getattr(np, 'asscalar', lambda a: a.item())
np.alen = getattr(np, 'alen', len)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import astropy.units as u                       # noqa: E402
from baseband import vdif                       # noqa: E402
import baseband                                 # noqa: E402
import bb_oracle_np as orc                      # noqa: E402

NFRAMES = 4000                  # 30.6 MiB of file, 128 M samples -- bench.py's sample
FRAME_RATE = 1000
PAYLOAD = 8000
SPF = 32000
REPEATS = 25


def make_file(path, seed=12345):
    """cfg2 layout written with plain NumPy (same bytes as
    baseband_amd.synth.random_vdif; no product code imported here)."""
    rng = np.random.default_rng(seed)
    frames = np.empty((NFRAMES, 8032), np.uint8)
    frames[:, 32:] = rng.integers(0, 256, size=(NFRAMES, PAYLOAD), dtype=np.uint8)
    w = frames[:, :32].view('<u4')
    k = np.arange(NFRAMES)
    # reference epoch 40 = 2020-01-01 (two per year since 2000)
    w[:, 0] = k // FRAME_RATE                                   # seconds, valid, not legacy
    w[:, 1] = (40 << 24) | (k % FRAME_RATE)                     # ref_epoch | frame_nr
    w[:, 2] = 8032 // 8                                         # vdif v0, 1 channel, frame_length
    w[:, 3] = (1 << 26) | (ord('A') << 8 | ord('A'))            # real, bps-1 = 1, thread 0, station
    w[:, 4:8] = 0
    frames.tofile(path)
    return frames.reshape(-1)


def pin_to_one_core():
    """Run on one core only (the reference is single-threaded; NumPy's take
    does not use threads): the last core this process may use."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, {cores[-1]})
        return cores[-1]
    except (AttributeError, OSError):
        return None


def stats(ts):
    q1, med, q3 = np.percentile(ts, [25, 50, 75])
    return {"median_s": round(float(med), 4), "q1_s": round(float(q1), 4), "q3_s": round(float(q3), 4),
            "iqr_over_median": round(float((q3 - q1) / med), 3), "min_s": round(float(np.min(ts)), 4),
            "runs_s": [round(float(t), 4) for t in ts]}


def main():
    tmp = tempfile.mkdtemp(prefix='bbcal_')
    path = os.path.join(tmp, 'cfg2.vdif')
    image = make_file(path)

    def ref_read(verify):
        with vdif.open(path, 'rs', sample_rate=SPF * FRAME_RATE * u.Hz, verify=verify) as fh:
            return fh.read()

    core = pin_to_one_core()
    fns = {'ref_v': lambda: ref_read(True), 'ref_n': lambda: ref_read(False),
           'port': lambda: orc.vdif_read(image, frame_rate=FRAME_RATE)[0]}
    outs, ts = {}, {k: [] for k in fns}
    for k, fn in fns.items():                                     # warm: imports, page cache, LUTs
        outs[k] = fn()
    for _ in range(REPEATS):
        for k, fn in fns.items():
            t0 = time.perf_counter()
            o = fn()
            ts[k].append(time.perf_counter() - t0)
            del o
    out_ref, out_ref_n, out_port = outs['ref_v'], outs['ref_n'], outs['port']
    same = (np.array_equal(np.ascontiguousarray(out_ref).view(np.uint32),
                           np.ascontiguousarray(out_port.reshape(out_ref.shape)).view(np.uint32))
            and np.array_equal(out_ref.view(np.uint32), out_ref_n.view(np.uint32)))
    nsamp = NFRAMES * SPF
    sv, sn, sp = stats(ts['ref_v']), stats(ts['ref_n']), stats(ts['port'])
    res = {
        "what": "same seeded cfg2 file ({} frames, {:.1f} MiB) decoded by the real reference "
                "(vdif.open(..,'rs').read()) and by oracle/bb_oracle_np.vdif_read: pinned to one core, "
                "{} rounds in which reference verify=True / verify=False / port take turns; "
                "medians and quartiles".format(NFRAMES, image.size / 2 ** 20, REPEATS),
        "reference": {"package": "baseband " + getattr(baseband, '__version__', '?'),
                      "verify_true": sv, "verify_false": sn,
                      "verify_true_Msps": round(nsamp / sv["median_s"] / 1e6, 1),
                      "verify_false_Msps": round(nsamp / sn["median_s"] / 1e6, 1)},
        "port": dict(sp, Msps=round(nsamp / sp["median_s"] / 1e6, 1)),
        "ratio_port_over_reference": round(sv["median_s"] / sp["median_s"], 3),
        "ratio_port_over_reference_verify_false": round(sn["median_s"] / sp["median_s"], 3),
        "ratio_from": "medians",
        "ratio_port_over_reference_quartile_range": [
            round(sv["q1_s"] / sp["q3_s"], 3), round(sv["q3_s"] / sp["q1_s"], 3)],
        "outputs_bit_identical": bool(same),
        "host": {"machine": platform.machine(), "cpus": os.cpu_count(), "pinned_to_core": core,
                 "python": platform.python_version(), "numpy": np.__version__},
        "note": "ratio = reference median seconds / port median seconds on THIS host; bench.py divides "
                "its port figure by it to estimate the reference-as-written rate on the GPU box's cores",
    }
    os.remove(path)
    os.rmdir(tmp)
    out = os.path.join(ROOT, 'tests', 'golden', 'cpu_calibration.json')
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
    if not same:
        raise SystemExit("reference and port outputs differ")


if __name__ == '__main__':
    main()
