"""Seeded random `subset` arguments -- integers, slices, lists, pairs of them, some out of range -- with and
without squeezing, on every format: sample shapes, samples, and what is refused."""
from ._dsl import *    # noqa: F401,F403

RATE_PH = (1e8 / 3) / 2 ** 23 * 2 ** 12 / 512
PH = [[S('gsb/sample_gsb_phased.Pol-L1.dat'), S('gsb/sample_gsb_phased.Pol-L2.dat')],
      [S('gsb/sample_gsb_phased.Pol-R1.dat'), S('gsb/sample_gsb_phased.Pol-R2.dat')]]
READERS = (
    ('vdif', [S('sample.vdif')], {}, (8,)),
    ('vdif', [S('sample_arochime.vdif')], dict(sample_rate=HZ(800e6 / 2 / 1024)), (2, 1024)),
    ('mark5b', [S('sample.m5b')], dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2), (8,)),
    ('mark4', [S('sample.m4')], dict(sample_rate=HZ(32e6), ntrack=64, decade=2010), (8,)),
    ('dada', [S('sample.dada')], {}, (2,)),
    ('guppi', [S('sample_puppi.raw')], {}, (2, 4)),
    ('gsb', [S('gsb/sample_gsb_phased.timestamp')], dict(raw=PH, sample_rate=HZ(RATE_PH), samples_per_frame=8), (2, 512)),
)


def one_index(x, n):
    kind = (x >> 20) % 6
    a, b = (x >> 24) % (n + 2), (x >> 34) % (n + 2)
    if kind == 0:
        return a - 1                                # (now and then -1 or n: the latter out of range)
    if kind == 1:
        return SL(min(a, b), max(a, b) + 1)
    if kind == 2:
        return SL(a % n, None, 1 + b % 3)
    if kind == 3:
        return sorted({a % n, b % n, (a + b) % n})
    if kind == 4:
        return [b % n, a % n] if a % n != b % n else [a % n]
    return SL(None)


def tries(k, fmt, args, kw, shape, n=12):
    steps, x = [], 4000 + k
    for j in range(n):
        parts = []
        for dim in shape:
            x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
            parts.append(one_index(x, dim))
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        if len(parts) == 2 and (x >> 30) % 3 == 0:
            parts = parts[:1]
        subset = parts[0] if len(parts) == 1 and (x >> 40) % 2 else TUP(*parts)
        squeeze = bool((x >> 45) % 2)
        steps += [open_('f', fmt, *args, 'rs', subset=subset, squeeze=squeeze, **kw), get('f.sample_shape'), get('f.shape'),
                  do('f.seek', 3), call(None, 'f.read', 4), close('f')]
    return steps


CASES = [
    case('subsets_drawn_at_random',
         'twelve random subsets for each of seven readers (threads, channels, polarisations; one and two '
         'axes; squeezed or not): the sample shape, four samples, or the refusal '
         '(the subset tests of base and of every format, arguments widened)',
         [tries(k, *r) for k, r in enumerate(READERS)]),
]
