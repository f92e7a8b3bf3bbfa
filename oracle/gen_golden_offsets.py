#!/opt/conda/bin/python3.9
"""Reference answers for the byte-slip table (`RawOffsets`, base/offsets.py:6-126;
filled by `_bad_frame`, base/base.py:1127-1219 and vdif/base.py:536-755):

* random sequences of assignments and look-ups on the reference's class;
* for the file-surgery cases already under tests/golden (vdif_corrupt_cases.json,
  fixed_corrupt_cases.json), the table the reference's reader is left with after a
  whole-file ``read()`` with verify='fix', and where it then puts every frame.

Run in the development container, next to the reference:

    /opt/conda/bin/python3.9 oracle/gen_golden_offsets.py

writes tests/golden/raw_offsets_cases.json (data only)."""
import io
import json
import os
import sys
import warnings

import numpy as np
np.asscalar = getattr(np, 'asscalar', lambda a: a.item())
np.alen = getattr(np, 'alen', len)
sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

import astropy.units as u                                   # noqa: E402
from astropy.time import Time                               # noqa: E402
from baseband import vdif, mark5b, mark4                    # noqa: E402
from baseband.base.offsets import RawOffsets                # noqa: E402

out = {"made_by": "oracle/gen_golden_offsets.py", "fuzz": [], "readers": {}}

# ---- the class by itself
rng = np.random.default_rng(20261003)
for trial in range(60):
    frame_nbytes = int(rng.choice([0, 1, 10, 1000, 5032]))
    ro = RawOffsets(frame_nbytes=frame_nbytes)
    steps = []
    pool = rng.integers(-20, 40, 4).tolist() + [0]
    for _ in range(int(rng.integers(1, 40))):
        fnr = int(rng.integers(0, 30))
        slip = int(rng.choice(pool))
        ro[fnr] = slip + fnr * frame_nbytes
        steps.append({"set": [fnr, slip + fnr * frame_nbytes], "frame_nr": list(ro.frame_nr),
                      "offset": list(ro.offset), "lookup": [int(ro[k]) for k in range(32)]})
    out["fuzz"].append({"frame_nbytes": frame_nbytes, "steps": steps, "repr": repr(ro)})


# ---- what the readers are left with
def table(fr, nframes):
    ro = fr._raw_offsets
    return {"frame_nr": [int(x) for x in ro.frame_nr], "offset": [int(x) for x in ro.offset],
            "frame_nbytes": int(ro.frame_nbytes), "lookup": [int(ro[k]) for k in range(nframes)]}


def read_all(opener, blob, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with opener(io.BytesIO(blob), 'rs', **kw) as fr:
            n = fr.shape[0] // fr.samples_per_frame
            fr.read()
            return table(fr, n)


with open(os.path.join(GOLD, 'vdif_corrupt_cases.json')) as f:
    vcases = json.load(f)
base = np.fromfile(os.path.join(GOLD, 'synth', 'vdif_triple.bin'), np.uint8)
lst = out["readers"]["vdif_triple"] = []
for c in vcases:
    keep = np.ones(len(base), bool)
    for lo, hi in c['remove']:
        keep[lo:hi] = False
    work = base.copy()
    for pos in c.get('flip', []):
        work[pos] ^= 0x55
    lst.append(read_all(vdif.open, work[keep].tobytes(), squeeze=False))

with open(os.path.join(GOLD, 'fixed_corrupt_cases.json')) as f:
    fixed = json.load(f)
files = np.load(os.path.join(GOLD, 'fixed_corrupt_files.npz'))
t0 = Time('2010-11-12T13:14:15')
openers = {
    'm5b_sample': (mark5b.open, dict(sample_rate=32 * u.MHz, kday=56000, nchan=8, bps=2)),
    'm5b_fake': (mark5b.open, dict(nchan=2, sample_rate=100 * u.kHz, ref_time=t0)),
    'm4_fake': (mark4.open, dict(sample_rate=100 * u.kHz, ref_time=t0)),
}
for group, cases in fixed.items():
    lst = out["readers"][group] = []
    for c in cases:
        if group == 'm5b_sample':
            b = np.fromfile(os.path.join(GOLD, 'samples', 'sample.m5b'), np.uint8).tobytes()
            tail = files['m5b_sample_tail'].tobytes()
        else:
            b, tail = files[group].tobytes(), b''
        lo, hi = c['remove']
        if c['kind'] == 'duplicate' or 'error' in c:
            lst.append(None)
            continue
        blob = b[:lo] + bytes.fromhex(c['replace']) + b[hi:] + tail
        opener, kw = openers[group]
        lst.append(read_all(opener, blob, **kw))

with open(os.path.join(GOLD, 'raw_offsets_cases.json'), 'w') as f:
    json.dump(out, f)
print('fuzz trials', len(out['fuzz']), {k: [None if t is None else (t['frame_nr'], t['offset']) for t in v]
                                       for k, v in out['readers'].items()})
