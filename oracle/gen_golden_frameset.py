#!/opt/conda/bin/python3.9
"""Differential fuzz fixture for VDIFFrameSet.fromfile: small
synthetic VDIF byte strings -- several threads in shuffled order, truncated
files, damaged headers, repeated threads, requests for thread subsets and for
threads that are missing -- read by the REAL reference
(baseband.vdif.frame.VDIFFrameSet.fromfile, vdif/frame.py:176-243).  Recorded
per case: the outcome (thread ids, header words and payload digest of every
frame, or the exception class) and the FILE POSITION the call leaves behind,
also when it raises.  Test infrastructure only; run in the development
container:   /opt/conda/bin/python3.9 -W ignore oracle/gen_golden_frameset.py
-> tests/golden/frameset_fuzz_cases.json
"""
import base64
import hashlib
import io
import json
import os
import sys

import numpy as np

np.asscalar = getattr(np, 'asscalar', lambda a: a.item())
np.alen = getattr(np, 'alen', len)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
from baseband.vdif.frame import VDIFFrameSet          # noqa: E402

PAYLOAD = 64


def frame(rng, frame_nr, thread, edv, legacy=False):
    hn = 16 if legacy else 32
    w = np.zeros(8, '<u4')
    w[0] = 100 | (int(legacy) << 30)
    w[1] = (40 << 24) | frame_nr
    w[2] = (hn + PAYLOAD) // 8                          # 1 channel, VDIF version 0
    w[3] = (1 << 26) | (thread << 16) | (ord('A') << 8 | ord('A'))      # 2 bits, real
    if not legacy and edv:
        w[4] = edv << 24
        w[5] = 0xACABFEED
    body = rng.integers(0, 256, PAYLOAD, dtype=np.uint8)
    return w[:hn // 4].tobytes() + body.tobytes()


def make_case(rng):
    nthread = int(rng.choice([1, 2, 3, 4, 8]))
    nsets = int(rng.integers(1, 4))
    edv = int(rng.choice([0, 0, 1]))
    legacy = bool(rng.random() < 0.15)
    fn = (16 if legacy else 32) + PAYLOAD
    frames = []
    for s in range(nsets):
        order = rng.permutation(nthread)
        for t in order:
            frames.append([s, int(t)])
    kind = rng.choice(['ok', 'truncate', 'damage', 'repeat', 'drop', 'ok'])
    if kind == 'repeat' and len(frames) > 1:
        k = int(rng.integers(1, len(frames)))
        frames.insert(k, list(frames[k - 1]))
    if kind == 'drop' and len(frames) > 1:
        del frames[int(rng.integers(0, len(frames)))]
    raw = bytearray(b''.join(frame(rng, s, t, edv, legacy) for s, t in frames))
    if kind == 'damage' and not legacy:
        k = int(rng.integers(0, len(frames)))
        off = k * fn + 20                                # word 5: the EDV sync, or must be zero for EDV 0
        raw[off:off + 4] = b'\\x11\\x22\\x33\\x44'
    if kind == 'truncate':
        raw = raw[:int(rng.integers(1, len(raw)))]
    start = 0 if rng.random() < 0.6 else fn * int(rng.integers(0, max(1, len(frames))))
    start = min(start, len(raw))
    choice = rng.random()
    if choice < 0.4:
        thread_ids = None
    elif choice < 0.8:
        thread_ids = sorted(int(t) for t in rng.choice(nthread, size=int(rng.integers(1, nthread + 1)), replace=False))
        if rng.random() < 0.5:
            thread_ids = [int(t) for t in rng.permutation(thread_ids)]
    else:
        thread_ids = [int(rng.integers(0, nthread)), nthread + 3]     # one that is not in the file
    verify = bool(rng.random() < 0.8)
    pass_edv = None if rng.random() < 0.5 else (False if legacy else edv)
    return bytes(raw), start, thread_ids, pass_edv, verify, kind


def run_reference(raw, start, thread_ids, edv, verify):
    fh = io.BytesIO(raw)
    fh.seek(start)
    try:
        fs = VDIFFrameSet.fromfile(fh, thread_ids=thread_ids, edv=edv, verify=verify)
    except Exception as exc:
        return {"raises": type(exc).__name__, "tell": fh.tell()}
    return {"tell": fh.tell(),
            "threads": [int(f.header['thread_id']) for f in fs.frames],
            "header0": [int(x) for x in fs.header0.words],
            "frames": [{"words": [int(x) for x in f.header.words],
                        "payload_sha": hashlib.sha256(np.asarray(f.payload.words).tobytes()).hexdigest()[:16]}
                       for f in fs.frames]}


def main():
    rng = np.random.default_rng(20260)
    cases = []
    while len(cases) < 400:
        raw, start, thread_ids, edv, verify, kind = make_case(rng)
        if len(raw) == 0:
            continue
        cases.append({"kind": str(kind), "raw": base64.b64encode(raw).decode(), "start": start,
                      "thread_ids": thread_ids, "edv": edv, "verify": verify,
                      "expect": run_reference(raw, start, thread_ids, edv, verify)})
    nraise = sum('raises' in c['expect'] for c in cases)
    out = os.path.join(ROOT, 'tests', 'golden', 'frameset_fuzz_cases.json')
    with open(out, 'w') as f:
        json.dump({"what": "VDIFFrameSet.fromfile of the reference on 400 drawn byte strings; "
                           "oracle/gen_golden_frameset.py", "cases": cases}, f)
    print(len(cases), "cases,", nraise, "raise;", os.path.getsize(out), "bytes")
    from collections import Counter
    print(Counter(c['expect'].get('raises', 'ok') for c in cases))


if __name__ == '__main__':
    main()
