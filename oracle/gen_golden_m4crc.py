"""Golden vectors for the Mark 4 longitudinal (along-track) header check.

Run in the development container with the interpreter that can import the
reference:   /opt/conda/bin/python3.9 -W ignore oracle/gen_golden_m4crc.py

BASELINE.json configs[3] names a "longitudinal-parity branch" for Mark 4.  What
the reference has along a track is the CRC-12 of its 160 header bits
(``baseband.mark4.header.crc12``: CRCStack(0x180f), base/utils.py:200-248; the
reference's own test asserts ``crc12.check(stream)`` on sample.m4,
mark4/tests/test_mark4.py:57-58).  The reference never applies it while
reading, so here it is an extra that flags frames and never alters samples
(SURVEY.md section 8a, row M4-x).  This script pins it: for every frame of the
sample files, of a seeded synthetic file written by the reference's writer, and
of copies with single header bits flipped, it records which tracks the
reference's ``crc12._crc(stream)`` finds non-zero -> tests/golden/mark4_crc_cases.json.
Only inputs (file name / seeded bit flips) and expected masks are stored.
"""
import io
import json
import os
import sys

import numpy as np

np.asscalar = getattr(np, 'asscalar', lambda a: a.item())
np.alen = getattr(np, 'alen', len)
sys.path.insert(0, '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

from baseband import mark4                                   # noqa: E402
from baseband.mark4.header import crc12                      # noqa: E402

DT = {16: '<u2', 32: '<u4', 64: '<u8'}


def bad_tracks(raw, offset0, ntrack, nframes):
    """Per frame: integer with bit k set when track k's 160 header bits do not
    divide by the CRC polynomial, straight from the reference's CRCStack."""
    fn = ntrack * 2500
    out = []
    for f in range(nframes):
        stream = np.frombuffer(raw, DT[ntrack], 160, offset0 + f * fn)
        rem = crc12._crc(stream)                # 12 words: remainder bits of every track
        out.append(int(np.bitwise_or.reduce(rem)))
    return out


def main():
    cases = []
    samples = [('samples/sample.m4', 64), ('samples/sample_32track.m4', 32),
               ('samples/sample_32track_fanout2.m4', 32), ('samples/sample_16track.m4', 16),
               ('samples/sample_64track_fanout2_ft.m4', 64), ('synth/m4_t64_f4.bin', 64),
               ('synth/m4_t16_f4.bin', 16)]
    rng = np.random.default_rng(160)
    for rel, ntrack in samples:
        raw = np.fromfile(os.path.join(GOLD, rel), np.uint8)
        with mark4.open(os.path.join(GOLD, rel), 'rb', ntrack=ntrack, decade=2010) as fh:
            fh.find_header()
            offset0 = fh.tell()
        fn = ntrack * 2500
        nframes = (len(raw) - offset0) // fn
        clean = bad_tracks(raw.tobytes(), offset0, ntrack, nframes)
        cases.append(dict(file=rel, ntrack=ntrack, offset0=offset0, nframes=nframes, flips=[],
                          bad=[format(b, 'x') for b in clean]))
        # flip single bits inside the headers (any of the 160 words x ntrack tracks)
        flips = []
        damaged = raw.copy()
        for _ in range(12):
            f = int(rng.integers(nframes))
            word = int(rng.integers(160))
            track = int(rng.integers(ntrack))
            byte = offset0 + f * fn + word * (ntrack // 8) + track // 8
            damaged[byte] ^= np.uint8(1 << (track % 8))
            flips.append([byte, track % 8])
        bad = bad_tracks(damaged.tobytes(), offset0, ntrack, nframes)
        cases.append(dict(file=rel, ntrack=ntrack, offset0=offset0, nframes=nframes, flips=flips,
                          bad=[format(b, 'x') for b in bad]))
        assert any(bad) or not flips
    with open(os.path.join(GOLD, 'mark4_crc_cases.json'), 'w') as f:
        json.dump(dict(reference='baseband.mark4.header.crc12 (CRCStack 0x180f), remainder over the '
                                 '160 header stream words of every frame', cases=cases), f, indent=0)
    print(len(cases), 'cases;', sum(len(c['bad']) for c in cases), 'frames')
    for c in cases:
        print(c['file'], c['ntrack'], 'flips', len(c['flips']), 'bad frames', sum(b != '0' for b in c['bad']))


if __name__ == '__main__':
    main()
