"""Streams with frames missing or damaged in the middle: what the readers fill in, warn about, or refuse."""
from ._dsl import *    # noqa: F401,F403

FB = 5032           # bytes of one frame of sample.vdif
M5B = 10016


def without(src, dst, nbytes, nframe, missing):
    """Steps that write `dst` = `src` minus the frames listed."""
    keep, parts = [i for i in range(nframe) if i not in missing], []
    runs, start = [], None
    for i in range(nframe + 1):
        if i in keep and start is None:
            start = i
        elif i not in keep and start is not None:
            runs.append((start, i))
            start = None
    steps = []
    for n, (a, b) in enumerate(runs):
        steps.append(fn('part%d' % n, 'file_bytes', src, a * nbytes, b * nbytes, quiet=True))
        parts.append(V('part%d' % n))
    steps.append(fn(None, 'write_file', dst, parts))
    return steps


SWEEP = ((27 * FB, 27 * FB + 1), (27 * FB + 3, 27 * FB + 4), (27 * FB + 4, 27 * FB + 8), (27 * FB + 13, 27 * FB + 14),
         (27 * FB + 31, 27 * FB + 32), (27 * FB + 20, 27 * FB + 40), (27 * FB + 32, 27 * FB + 33),
         (27 * FB + 5031, 27 * FB + 5032), (27 * FB + 5030, 27 * FB + 5034), (27 * FB + 4000, 28 * FB + 4000),
         (27 * FB + 100, 28 * FB + 200), (27 * FB + 100, 29 * FB + 50), (30 * FB + 7, 33 * FB + 7),
         (26 * FB + 2500, 36 * FB + 2400), (33 * FB + 16, 33 * FB + 17), (39 * FB + 1000, 39 * FB + 1016),
         (34 * FB + 8, 34 * FB + 12), (25 * FB + 5031, 26 * FB + 1))

M5B_SWEEP = ((5 * M5B, 5 * M5B + 1), (5 * M5B + 2, 5 * M5B + 4), (5 * M5B + 4, 5 * M5B + 8), (5 * M5B + 9, 5 * M5B + 10),
             (5 * M5B + 14, 5 * M5B + 16), (5 * M5B + 16, 5 * M5B + 17), (5 * M5B + 10015, 5 * M5B + 10016),
             (5 * M5B + 10010, 5 * M5B + 10020), (5 * M5B + 5000, 6 * M5B + 5000), (5 * M5B + 100, 6 * M5B + 200),
             (5 * M5B + 100, 7 * M5B + 50), (3 * M5B + 7, 6 * M5B + 7), (8 * M5B + 4000, 8 * M5B + 4100),
             (10 * M5B + 16, 10 * M5B + 20), (11 * M5B + 500, 11 * M5B + 504), (1 * M5B + 20, 1 * M5B + 24),
             (0 * M5B + 5000, 0 * M5B + 5004))

M4B = 160000
M4_SWEEP = ((3 * M4B, 3 * M4B + 8), (3 * M4B + 500, 3 * M4B + 508), (3 * M4B + 640, 3 * M4B + 1280),
            (3 * M4B + 5000, 3 * M4B + 5016), (3 * M4B + 159992, 3 * M4B + 160008), (3 * M4B + 80000, 4 * M4B + 80000),
            (2 * M4B + 100, 4 * M4B + 50), (6 * M4B + 1000, 6 * M4B + 1008), (7 * M4B + 3000, 7 * M4B + 3008),
            (1 * M4B + 768, 1 * M4B + 776))

def _drawn(n, seed, nbytes=FB, first=26, span=20, nframe=48):
    out, x = [], seed
    for _ in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        start = first * nbytes + (x >> 20) % (span * nbytes)
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        kind = (x >> 33) % 4
        length = (1, 1 + (x >> 40) % 64, 1 + (x >> 40) % nbytes, nbytes + (x >> 40) % (3 * nbytes // 2))[kind]
        out.append((start, min(start + length, nframe * nbytes)))
    return tuple(out)


RANDOM_LOSSES = _drawn(40, 20261003)


def _frames_drawn(n, seed, first, stop, most=6):
    out, x = [], seed
    for _ in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        count = 1 + (x >> 30) % most
        gone = set()
        while len(gone) < count:
            x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
            gone.add(first + (x >> 25) % (stop - first))
        out.append(sorted(gone))
    return out

GIVES_UP = (34 * FB + 8, 34 * FB + 12)   # word 2 of a header: the reference finds no header nearby and raises

VDIF_MISSING = ([5], [8], [15], [47], [7, 8], [8, 9, 10, 11, 12, 13, 14, 15], [10, 11, 30], [16, 17, 18, 19, 20, 21, 22, 23, 24],
                [41, 42, 43, 44, 45, 46, 47])

CASES = [
    case('vdif_frames_missing',
         'six frame sets of eight threads with frames taken out -- one thread of a set, the last of one set '
         'and the first of the next, a whole set, the very last frame, the tail: by default the holes are '
         'filled with the fill value and reported (for the file that lacks set 2 and the first frame of set '
         '3 the intact set 1 is named too, without more: the read ahead of it fails); verify=True refuses '
         'where the reference does '
         '(vdif/tests/test_vdif.py, TestCorruptSampleCopy / test_missing_frames)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         digest(T('base.vdif')),
         [without(T('base.vdif'), T('m%d.vdif' % k), FB, 48, miss)
          + [open_('f', 'vdif', T('m%d.vdif' % k), 'rs'), get('f.shape'), get('f.stop_time'), get('f.verify'),
             call(None, 'f.read', some_warns=True), do('f.seek', 19990),
             call(None, 'f.read', 20, some_warns=True),
             get('f.info.errors'), get('f.info.warnings'), close('f'),
             open_('g', 'vdif', T('m%d.vdif' % k), 'rs', verify=True), get('g.shape'),
             call(None, 'g.read', some_warns=True), close('g'),
             open_('b', 'vdif', T('m%d.vdif' % k), 'rb'), get('b.info.number_of_frames'),
             get('b.info.number_of_framesets'), close('b')]
          for k, miss in enumerate(VDIF_MISSING)]),

    case('vdif_frames_missing_and_a_subset',
         'a reader of some threads only meets only the holes in those: threads 1 and 5 of files that lack '
         'thread 5 of set 0, thread 0 of set 1, a whole set; the pieces read into a caller\'s array; a strict '
         'reader of thread 3 alone reads what is whole for it.  (Behind a hole the reference, reading some '
         'threads only, warns "problem loading frame set k." for every other set, without saying what about; '
         'this package names the sets that lack a thread that was asked for -- warnings not compared here) '
         '(vdif/tests/test_vdif.py, missing frames with subset)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [without(T('base.vdif'), T('m%d.vdif' % k), FB, 48, miss)
          + [open_('f', 'vdif', T('m%d.vdif' % k), 'rs', subset=[1, 5]), get('f.shape'), get('f.sample_shape'),
             call(None, 'f.read', any_warns=True), do('f.seek', 19990), let('o', ZEROS((20, 2), 'f4')),
             call(None, 'f.read', out=V('o'), quiet=True, any_warns=True), get('o'), close('f'),
             open_('g', 'vdif', T('m%d.vdif' % k), 'rs', subset=TUP(3), verify=True), get('g.shape'),
             call(None, 'g.read', any_warns=True), close('g')]
          for k, miss in enumerate(([5], [8], [8, 9, 10, 11, 12, 13, 14, 15], [11, 21]))]),

    case('vdif_bytes_missing',
         'bytes lost inside a payload and inside a header, in the second half of the file (with a loss in the '
         'first three sets the reference does not open the file at all, and after one in the first word of a '
         'header it finds no header nearby): the frames after the loss sit at odd '
         'offsets; by default they are found again, verify=True stops at the header that is none '
         '(vdif/tests/test_vdif.py, TestCorruptSampleCopy / test_missing_bytes)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [[fn('a', 'file_bytes', T('base.vdif'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.vdif'), hi, None, quiet=True),
           fn(None, 'write_file', T('b%d.vdif' % k), [V('a'), V('b')]),
           open_('f', 'vdif', T('b%d.vdif' % k), 'rs'), get('f.shape'),
           call(None, 'f.read', some_warns=True), get('f.info.checks'), close('f'),
           open_('g', 'vdif', T('b%d.vdif' % k), 'rs', verify=True), call(None, 'g.read', any_warns=True),
           get('g.info.checks'), close('g')]
          for k, (lo, hi) in enumerate(((31 * FB + 10, 31 * FB + 20), (40 * FB + 32, 40 * FB + 33),
                                        (29 * FB + 5000, 29 * FB + 5040), (37 * FB + 2000, 37 * FB + 2008)))]),

    case('vdif_byte_losses_swept',
         'the same file with bytes lost at a sweep of places in its second half: at the first, a middle and '
         'the last byte of a header, across a header, at both ends of a payload, over more than a frame, over '
         'more than a frame set; only the default (repairing) read; where word 2 of a header is lost the '
         'reference finds no header nearby and raises, this package reads on (we_may_manage) (vdif/tests/test_vdif.py, '
         'TestCorruptSampleCopy / test_missing_bytes, positions widened)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [[fn('a', 'file_bytes', T('base.vdif'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.vdif'), hi, None, quiet=True),
           fn(None, 'write_file', T('s%d.vdif' % k), [V('a'), V('b')]),
           open_('f', 'vdif', T('s%d.vdif' % k), 'rs'), get('f.shape'),
           call(None, 'f.read', **({'some_warns': True} if (lo, hi) != GIVES_UP else
                                   {'any_warns': True, 'we_may_manage': True})), close('f')]
          for k, (lo, hi) in enumerate(SWEEP)]),

    case('vdif_losses_drawn_at_random',
         'forty more losses in the second half of the same file, places and lengths drawn from a seeded '
         'generator (1 byte to two and a half frames): shapes and repaired samples '
         '(vdif/tests/test_vdif.py, TestCorruptSampleCopy, widened)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [[fn('a', 'file_bytes', T('base.vdif'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.vdif'), hi, None, quiet=True),
           fn(None, 'write_file', T('r%d.vdif' % k), [V('a'), V('b')]),
           open_('f', 'vdif', T('r%d.vdif' % k), 'rs', we_may_manage=True, quiet=True),
           call(None, 'f.read', any_warns=True, we_may_manage=True), close('f')]
          for k, (lo, hi) in enumerate(RANDOM_LOSSES)]),

    case('frames_missing_drawn_at_random',
         'one to six whole frames taken out of the second half of the VDIF file (thirty draws) and of the '
         'Mark 5B file (twenty): shapes and repaired samples (the missing-frame tests of both formats, widened)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [without(T('base.vdif'), T('q%d.vdif' % k), FB, 48, miss)
          + [open_('f', 'vdif', T('q%d.vdif' % k), 'rs', we_may_manage=True, quiet=True), get('f.shape'),
             call(None, 'f.read', any_warns=True, we_may_manage=True), close('f')]
          for k, miss in enumerate(_frames_drawn(30, 5, 24, 48))],
         open_('fr', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         call('d', 'fr.read'),
         open_('fw', 'mark5b', T('base.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [without(T('base.m5b'), T('q%d.m5b' % k), M5B, 12, miss)
          + [open_('f', 'mark5b', T('q%d.m5b' % k), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2,
                   we_may_manage=True, quiet=True), get('f.shape'),
             call(None, 'f.read', any_warns=True, we_may_manage=True), close('f')]
          for k, miss in enumerate(_frames_drawn(20, 6, 1, 12, most=3))]),

    case('damage_inside_a_sequence_of_files',
         'the six frame sets split over three files by a template; a frame, forty bytes, a whole set taken out '
         'of the MIDDLE file: the stream is read through the template and repaired across the file boundaries '
         '(vdif template tests and the corrupt-file tests, combined)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('p{file_nr:d}.vdif'), 'ws', header0=V('fr.header0'), nthread=8, file_size=16 * FB),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'), listdir(),
         fn('mid', 'file_bytes', T('p1.vdif'), quiet=True),
         [[fn(None, 'write_file', T('p1.vdif'), [V('mid')]),
           fn('a', 'file_bytes', T('p1.vdif'), 0, lo, quiet=True), fn('b', 'file_bytes', T('p1.vdif'), hi, None, quiet=True),
           fn(None, 'write_file', T('p1.vdif'), [V('a'), V('b')]),
           open_('f', 'vdif', T('p{file_nr:d}.vdif'), 'rs', we_may_manage=True, quiet=True), get('f.shape'),
           call(None, 'f.read', some_warns=True, we_may_manage=True), do('f.seek', 59990),
           call(None, 'f.read', 20, any_warns=True, we_may_manage=True), close('f'),
           open_('g', 'vdif', [T('p0.vdif'), T('p1.vdif'), T('p2.vdif')], 'rs', verify=True),
           call(None, 'g.read', any_warns=True), close('g')]
          for lo, hi in ((5 * FB, 6 * FB), (9 * FB + 3000, 9 * FB + 3040), (8 * FB, 16 * FB), (15 * FB, 16 * FB))]),

    case('info_of_damaged_files',
         'what `info` says of damaged files -- readable or not, the verdict on continuity and where the '
         'first trouble is -- for repairing and for strict readers of VDIF, Mark 5B and Mark 4 files with '
         'frames or bytes missing (the info checks of the corrupt-file tests of the three formats)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [[fn('a', 'file_bytes', T('base.vdif'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.vdif'), hi, None, quiet=True),
           fn(None, 'write_file', T('i%d.vdif' % k), [V('a'), V('b')]),
           [[open_('f', 'vdif', T('i%d.vdif' % k), 'rs', verify=v), get('f.info.readable'), get('f.info.checks'),
             get('f.info.errors', prefix=24), get('f.info.warnings', prefix=24), call(None, 'f.tell'), close('f')]
            for v in ('fix', True)]]
          for k, (lo, hi) in enumerate(((29 * FB, 30 * FB), (32 * FB, 40 * FB), (47 * FB, 48 * FB), (31 * FB + 10, 31 * FB + 20),
                                        (37 * FB + 2000, 37 * FB + 2008), (44 * FB + 100, 45 * FB + 300)))],
         open_('fr', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         call('d', 'fr.read'),
         open_('fw', 'mark5b', T('base.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [[fn('a', 'file_bytes', T('base.m5b'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.m5b'), hi, None, quiet=True),
           fn(None, 'write_file', T('i%d.m5b' % k), [V('a'), V('b')]),
           [[open_('f', 'mark5b', T('i%d.m5b' % k), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2, verify=v),
             get('f.info.readable'), get('f.info.checks'), get('f.info.errors', prefix=24), get('f.info.warnings', prefix=24),
             close('f')] for v in ('fix', True)]]
          for k, (lo, hi) in enumerate(((5 * M5B, 6 * M5B), (3 * M5B, 5 * M5B), (11 * M5B, 12 * M5B), (7 * M5B + 20, 7 * M5B + 24),
                                        (9 * M5B + 5000, 9 * M5B + 5010)))],
         open_('fr', 'mark4', S('sample.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010),
         call('d', 'fr.read'),
         open_('fw', 'mark4', T('base.m4'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6)),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'),
         close('fr'),
         [[fn('a', 'file_bytes', T('base.m4'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.m4'), hi, None, quiet=True),
           fn(None, 'write_file', T('i%d.m4' % k), [V('a'), V('b')]),
           [[open_('f', 'mark4', T('i%d.m4' % k), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010, verify=v),
             get('f.info.readable'), get('f.info.checks'), get('f.info.errors', prefix=24), get('f.info.warnings', prefix=24),
             close('f')] for v in ('fix', True)]]
          for k, (lo, hi) in enumerate(((3 * M4B, 4 * M4B), (2 * M4B, 4 * M4B), (5 * M4B + 5000, 5 * M4B + 5016)))]),

    case('mark5b_and_mark4_losses_drawn_at_random',
         'the same for twelve Mark 5B frames (forty losses) and eight Mark 4 frames (twenty): shapes and '
         'repaired samples (the corrupt-stream tests of both formats, widened)',
         open_('fr', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         call('d', 'fr.read'),
         open_('fw', 'mark5b', T('base.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [[fn('a', 'file_bytes', T('base.m5b'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.m5b'), hi, None, quiet=True),
           fn(None, 'write_file', T('r%d.m5b' % k), [V('a'), V('b')]),
           open_('f', 'mark5b', T('r%d.m5b' % k), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2,
                 we_may_manage=True, quiet=True),
           call(None, 'f.read', any_warns=True, we_may_manage=True), close('f')]
          for k, (lo, hi) in enumerate(_drawn(40, 77, M5B, 1, 11, 12))],
         open_('fr', 'mark4', S('sample.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010),
         call('d', 'fr.read'),
         open_('fw', 'mark4', T('base.m4'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6)),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'),
         close('fr'),
         [[fn('a', 'file_bytes', T('base.m4'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.m4'), hi, None, quiet=True),
           fn(None, 'write_file', T('r%d.m4' % k), [V('a'), V('b')]),
           open_('f', 'mark4', T('r%d.m4' % k), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010,
                 we_may_manage=True, quiet=True),
           call(None, 'f.read', any_warns=True, we_may_manage=True), close('f')]
          for k, (lo, hi) in enumerate(_drawn(20, 78, M4B, 1, 7, 8))]),

    case('mark4_narrow_layouts_losses_drawn_at_random',
         'the same for Mark 4 files of 32 tracks (fan-out 2) and of 16 tracks, eight frames each written from the '
         'two of their sample, twelve losses each: the header search on 32- and 16-bit stream words '
         '(mark4/tests/test_mark4.py corrupt-stream tests, layouts widened)',
         [[open_('fr', 'mark4', S(sample), 'rs', sample_rate=HZ(rate), ntrack=ntrack, decade=2010), call('d', 'fr.read'),
           open_('fw', 'mark4', T('base%d.m4' % ntrack), 'ws', header0=V('fr.header0'), sample_rate=HZ(rate)),
           do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'),
           close('fr'), digest(T('base%d.m4' % ntrack)),
           [[fn('a', 'file_bytes', T('base%d.m4' % ntrack), 0, lo, quiet=True),
             fn('b', 'file_bytes', T('base%d.m4' % ntrack), hi, None, quiet=True),
             fn(None, 'write_file', T('r%d_%d.m4' % (ntrack, k)), [V('a'), V('b')]),
             open_('f', 'mark4', T('r%d_%d.m4' % (ntrack, k)), 'rs', sample_rate=HZ(rate), ntrack=ntrack, decade=2010,
                   we_may_manage=True, quiet=True),
             call(None, 'f.read', any_warns=True, we_may_manage=True), close('f')]
            for k, (lo, hi) in enumerate(_drawn(12, 80 + ntrack, ntrack * 2500, 1, 7, 8))]]
          for sample, ntrack, rate in (('sample_32track_fanout2.m4', 32, 16e6), ('sample_16track.m4', 16, 32e6))]),

    case('vdif_headers_damaged_in_place',
         'no bytes lost, but a header overwritten -- the sync pattern of one frame, the frame length of another, '
         'two headers next to each other: by default the frame (or what the search cannot reach behind '
         'it) is filled, verify=True stops at the header; without verification the reference trips over '
         'the zeroed frame length (a negative read length), this package reads by position (marked '
         'we_may_manage) (vdif/tests/test_vdif.py, TestCorruptSampleCopy / test_bad_header)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         fn('all', 'file_bytes', T('base.vdif'), quiet=True),
         [[fn(None, 'write_file', T('p%d.vdif' % k), [V('all')])]
          + [fn(None, 'patch_file', T('p%d.vdif' % k), pos, HEX(val)) for pos, val in patches]
          + [open_('f', 'vdif', T('p%d.vdif' % k), 'rs'), get('f.shape'),
             call(None, 'f.read', some_warns=True), get('f.info.checks'), close('f'),
             open_('g', 'vdif', T('p%d.vdif' % k), 'rs', verify=True), call(None, 'g.read', any_warns=True), close('g'),
             open_('u', 'vdif', T('p%d.vdif' % k), 'rs', verify=False),
             call(None, 'u.read', any_warns=True, we_may_manage=True, quiet=(k == 1) or None), close('u')]
          for k, patches in enumerate((((27 * FB + 20, 'ffffffff'),), ((28 * FB + 8, '00000000'),),
                                       ((29 * FB + 20, '00'), (30 * FB + 23, '00')), ((35 * FB + 1, 'ff'),)))]),

    case('vdif_other_fill_value',
         'the fill value of a reader is what holes and invalid frames decode to (base/tests/test_base.py '
         'fill_value; vdif invalid frames)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('base.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         do('fw.write', V('d')), do('fw.write', V('d'), valid=False), do('fw.write', V('d')), close('fw'), close('fr'),
         without(T('base.vdif'), T('m.vdif'), FB, 48, [3, 4, 44]),
         [[open_('f', 'vdif', T('m.vdif'), 'rs', fill_value=fv), get('f.fill_value'), call('x', 'f.read', some_warns=True),
           item(None, 'x', TUP(SL(0, 2), SL(None))), item(None, 'x', TUP(SL(40000, 40002), SL(None))),
           item(None, 'x', TUP(SL(100000, 100002), SL(None))), close('f')]
          for fv in (0.0, -999.0, 7.5)]),

    case('mark5b_frames_missing',
         'twelve Mark 5B frames with frames taken out of the middle, the start and the end: gaps are filled '
         'and frames are found again by their sync words and counters; where four frames in a row are '
         'gone the reference looks for frame 8 at the end of the file, finds no header nearby and gives '
         'up -- this package places the frames it located and reads on (marked we_may_manage) '
         '(mark5b/tests/test_mark5b.py, TestCorruptStream / test_missing_frames)',
         open_('fr', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         call('d', 'fr.read'),
         open_('fw', 'mark5b', T('base.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         digest(T('base.m5b')),
         [without(T('base.m5b'), T('m%d.m5b' % k), M5B, 12, miss)
          + [open_('f', 'mark5b', T('m%d.m5b' % k), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
             get('f.shape'), get('f.start_time'), get('f.stop_time'),
             call(None, 'f.read', some_warns=True, we_may_manage=True),
             do('f.seek', 4990), call(None, 'f.read', 20, some_warns=True), close('f'),
             open_('g', 'mark5b', T('m%d.m5b' % k), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2, verify=True),
             call(None, 'g.read', some_warns=True), close('g')]
          for k, miss in enumerate(([1], [3, 4], [11], [0], [5, 6, 7, 8], [2, 9]))]),

    case('mark5b_byte_losses_swept',
         'twelve Mark 5B frames with bytes lost at a sweep of places: in the sync word, in the time code, in '
         'its CRC, at both ends of a payload, over more than a frame; the default (repairing) read '
         '(mark5b/tests/test_mark5b.py, corrupt stream cases, positions widened)',
         open_('fr', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         call('d', 'fr.read'),
         open_('fw', 'mark5b', T('base.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         [[fn('a', 'file_bytes', T('base.m5b'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.m5b'), hi, None, quiet=True),
           fn(None, 'write_file', T('s%d.m5b' % k), [V('a'), V('b')]),
           open_('f', 'mark5b', T('s%d.m5b' % k), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2), get('f.shape'),
           call(None, 'f.read', some_warns=True), close('f')]
          for k, (lo, hi) in enumerate(M5B_SWEEP)]),

    case('mark4_frames_and_bytes_missing',
         'eight Mark 4 frames (64 tracks, written from the two of the sample) with whole frames and bytes '
         'taken out: gaps filled, frames found again by the sync pattern in every track '
         '(mark4/tests/test_mark4.py, test_corrupt_stream cases, positions widened)',
         open_('fr', 'mark4', S('sample.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010),
         call('d', 'fr.read'),
         open_('fw', 'mark4', T('base.m4'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6)),
         do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), do('fw.write', V('d')), close('fw'),
         close('fr'), digest(T('base.m4')),
         [without(T('base.m4'), T('m%d.m4' % k), M4B, 8, miss)
          + [open_('f', 'mark4', T('m%d.m4' % k), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010),
             get('f.shape'), get('f.start_time'), get('f.stop_time'),
             call(None, 'f.read', some_warns=True, we_may_manage=True), close('f'),
             open_('g', 'mark4', T('m%d.m4' % k), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010, verify=True),
             call(None, 'g.read', any_warns=True), close('g')]
          for k, miss in enumerate(([1], [3, 4], [7], [0], [2, 5]))],
         [[fn('a', 'file_bytes', T('base.m4'), 0, lo, quiet=True), fn('b', 'file_bytes', T('base.m4'), hi, None, quiet=True),
           fn(None, 'write_file', T('s%d.m4' % k), [V('a'), V('b')]),
           open_('f', 'mark4', T('s%d.m4' % k), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010), get('f.shape'),
           call(None, 'f.read', some_warns=True, we_may_manage=True), close('f')]
          for k, (lo, hi) in enumerate(M4_SWEEP)]),

    case('mark5b_damage_inside_frames',
         'bytes lost inside a frame (the file is shorter by less than a frame) and a sync word overwritten '
         'in the middle of a file (test_mark5b.py, corrupt stream cases with partial frames)',
         open_('fr', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         call('d', 'fr.read'),
         open_('fw', 'mark5b', T('base.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         do('fw.write', V('d')), do('fw.write', V('d')), close('fw'), close('fr'),
         fn('a', 'file_bytes', T('base.m5b'), 0, 3 * M5B + 2000, quiet=True),
         fn('b', 'file_bytes', T('base.m5b'), 3 * M5B + 2400, None, quiet=True),
         fn(None, 'write_file', T('short.m5b'), [V('a'), V('b')]),
         open_('f', 'mark5b', T('short.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         get('f.shape'), call('x', 'f.read', some_warns=True), item(None, 'x', SL(14990, 15010)),
         item(None, 'x', SL(19990, 20010)), close('f'),
         fn('all', 'file_bytes', T('base.m5b'), quiet=True), fn(None, 'write_file', T('nosync.m5b'), [V('all')]),
         fn(None, 'patch_file', T('nosync.m5b'), 5 * M5B, HEX('00000000')),
         open_('f2', 'mark5b', T('nosync.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         get('f2.shape'), call('y', 'f2.read', some_warns=True), item(None, 'y', SL(24990, 25010)),
         item(None, 'y', SL(29990, 30010)), close('f2')),
]
