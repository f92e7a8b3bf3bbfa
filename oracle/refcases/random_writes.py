"""Stream writers of every format fed pieces of seeded random lengths, some flagged invalid: the bytes on disk."""
from ._dsl import *    # noqa: F401,F403
from .writers import pieces, L2, L8


def cuts_and_flags(seed, total, npiece):
    x, cuts = seed, {0, total}
    while len(cuts) < npiece + 1:
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        cuts.add((x >> 21) % total)
    cuts = sorted(cuts)
    flags = []
    for _ in range(len(cuts) - 1):
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        flags.append((x >> 40) % 4 != 0)
    return cuts, flags


def written(k, fmt, name, shape, levels, cplx, kw, read_kw, npiece=9):
    c, f = cuts_and_flags(300 + k, shape[0], npiece)
    return [let('d', RNG(900 + k, shape, levels, complex=cplx)),
            open_('fw', fmt, T(name), 'ws', squeeze=False, **kw),
            pieces('d', c, f), call(None, 'fw.tell'), close('fw'), digest(T(name)),
            open_('fr', fmt, T(name), 'rs', squeeze=False, **read_kw), get('fr.shape'), call(None, 'fr.read'), close('fr')]


def sequence(k, fmt, template, shape, levels, cplx, kw, read_kw, file_size, length, nread=10, names=None):
    c, f = cuts_and_flags(400 + k, shape[0], 6)
    steps = [let('d', RNG(950 + k, shape, levels, complex=cplx)),
             open_('fw', fmt, T(template), 'ws', squeeze=False, file_size=file_size, **kw),
             pieces('d', c, [True] * len(f)), close('fw'), listdir(),
             open_('fr', fmt, T(template) if names is None else [T(n) for n in names], 'rs', squeeze=False, **read_kw),
             get('fr.shape')]
    x = 500 + k
    for _ in range(nread):
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        offset = (x >> 24) % length
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        count = 1 + (x >> 28) % max(1, min(length - offset, length // 2))
        steps += [do('fr.seek', offset), call(None, 'fr.read', count)]
    steps += [do('fr.seek', 0), call(None, 'fr.read'), close('fr')]
    return steps


CASES = [
    case('pieces_of_random_length',
         'VDIF (2-bit 8 threads, 4-bit complex 2 threads x 4 channels, 8-bit), Mark 5B, Mark 4 (two layouts), '
         'DADA and GUPPI writers, each fed nine or so pieces whose lengths a seeded generator drew, a quarter of '
         'them flagged invalid: file digests and what a reader returns (the stream writer tests of every '
         'format, piece sizes widened)',
         written(0, 'vdif', 'a.vdif', (6144, 8, 1), L2, False,
                 dict(sample_rate=HZ(2.048e6), nthread=8, nchan=1, bps=2, complex_data=False, edv=1, station='rw',
                      samples_per_frame=1024, time=TIME('2021-03-04T05:06:07')), {}),
         written(1, 'vdif', 'b.vdif', (2400, 2, 4), [-2.0, -1.0, -0.2, 0.0, 0.3, 1.0, 2.0], True,
                 dict(sample_rate=HZ(48000.), nthread=2, nchan=4, bps=4, complex_data=True, edv=0, station='rw',
                      samples_per_frame=240, time=TIME('2021-03-04T05:06:07')), dict(sample_rate=HZ(48000.))),
         written(2, 'vdif', 'c.vdif', (1536, 1, 2), L8, False,
                 dict(sample_rate=HZ(128000.), nthread=1, nchan=2, bps=8, complex_data=False, edv=False, station='rw',
                      samples_per_frame=128, time=TIME('2021-03-04T05:06:07')), dict(sample_rate=HZ(128000.))),
         written(3, 'mark5b', 'd.m5b', (30000, 8), L2, False,
                 dict(sample_rate=HZ(32e6), nchan=8, bps=2, time=TIME('2014-06-13T05:30:01')),
                 dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2)),
         written(4, 'mark5b', 'e.m5b', (4 * 40000, 2), [-1.0, 1.0], False,
                 dict(sample_rate=HZ(32e6), nchan=2, bps=1, time=TIME('2014-06-13T23:59:59.99')),
                 dict(sample_rate=HZ(32e6), kday=56000, nchan=2, bps=1)),
         written(5, 'mark4', 'f.m4', (4 * 80000, 8), L2, False,
                 dict(sample_rate=HZ(32e6), ntrack=64, fanout=4, nchan=8, bps=2, time=TIME('2014-06-16T07:38:12.47500')),
                 dict(sample_rate=HZ(32e6), ntrack=64, decade=2010), npiece=7),
         written(6, 'mark4', 'g.m4', (4 * 40000, 8), L2, False,
                 dict(sample_rate=HZ(16e6), ntrack=32, fanout=2, nchan=8, bps=2, time=TIME('2014-06-16T07:38:12.47500')),
                 dict(sample_rate=HZ(16e6), ntrack=32, decade=2010), npiece=7),
         written(7, 'dada', 'h.dada', (5000, 2, 1), L8, True,
                 dict(time=TIME('2013-07-02T01:39:20'), sample_rate=HZ(16e6), samples_per_frame=1000, npol=2, nchan=1,
                      bps=8, complex_data=True), {}),
         open_('fg', 'guppi', S('sample_puppi.raw'), 'rs'), call('hg', 'fg.header0.copy'), close('fg'),
         set_('hg.overlap', 0), set_('hg.samples_per_frame', 512),
         written(8, 'guppi', 'i.raw', (2560, 2, 4), L8, True, dict(header0=V('hg')), {})),

    case('sequences_of_files',
         'streams split over files by a name template and a file size -- VDIF (three frame sets per file), '
         'Mark 5B (two frames; read back by the list of names), Mark 4 (one frame), DADA (one block per file, its own naming) -- then read '
         'back through the template at ten random places and whole: the names and sizes of the files, the '
         'samples across their boundaries (the template tests of every format, sizes and places widened)',
         sequence(0, 'vdif', 'a{file_nr:02d}.vdif', (16 * 512, 4, 2), L2, False,
                  dict(sample_rate=HZ(512e3), nthread=4, nchan=2, bps=2, complex_data=False, edv=1, station='sq',
                       samples_per_frame=512, time=TIME('2021-03-04T05:06:07')), {}, 3 * 4 * (32 + 256), 16 * 512),
         sequence(1, 'mark5b', 'b{file_nr:d}.m5b', (9 * 5000, 8), L2, False,
                  dict(sample_rate=HZ(32e6), nchan=8, bps=2, time=TIME('2014-06-13T05:30:01')),
                  dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2), 2 * 10016, 9 * 5000,
                  names=['b%d.m5b' % j for j in range(5)]),     # (a template would lose `kday` in the reference)
         sequence(2, 'mark4', 'c{file_nr:d}.m4', (3 * 80000, 8), L2, False,
                  dict(sample_rate=HZ(32e6), ntrack=64, fanout=4, nchan=8, bps=2, time=TIME('2014-06-16T07:38:12.47500')),
                  dict(sample_rate=HZ(32e6), ntrack=64, decade=2010), 160000, 3 * 80000),
         let('dd', RNG(990, (4000, 2, 1), L8, complex=True)),
         open_('fw', 'dada', T('{utc_start}.{obs_offset:016d}.000000.dada'), 'ws', squeeze=False,
               time=TIME('2013-07-02T01:39:20'), sample_rate=HZ(16e6), samples_per_frame=1000, npol=2, nchan=1, bps=8,
               complex_data=True),
         do('fw.write', V('dd')), close('fw'), listdir(),
         open_('fr', 'dada', T('2013-07-02-01:39:20.{obs_offset:016d}.000000.dada'), 'rs', squeeze=False), get('fr.shape'),
         [[do('fr.seek', o), call(None, 'fr.read', n)] for o, n in ((990, 20), (1999, 2), (2500, 1500), (3999, 1))],
         close('fr')),
]
