"""Seeded random walks of seek and read over every sample recording, whole and with subsets: the samples
returned are the reference's at every step."""
from ._dsl import *    # noqa: F401,F403

GSB_RATE = (1e8 / 3) / 2 ** 23
WALKS = (
    ('vdif', [S('sample.vdif')], {}, 40000),
    ('vdif', [S('sample.vdif')], dict(subset=[1, 6, 3]), 40000),
    ('vdif', [S('sample.vdif')], dict(squeeze=False, subset=TUP(SL(2, 7, 2), 0)), 40000),
    ('vdif', [S('sample_vlbi.vdif')], {}, 40000),
    ('vdif', [S('sample_mwa.vdif')], dict(sample_rate=HZ(1.28e6)), 1280),
    ('vdif', [S('sample_arochime.vdif')], dict(sample_rate=HZ(800e6 / 2 / 1024), subset=TUP(1, SL(100, 200))), 5),
    ('vdif', [S('sample_bps1.vdif')], dict(sample_rate=HZ(8e6)), 8000),
    ('mark5b', [S('sample.m5b')], dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2), 20000),
    ('mark5b', [S('sample.m5b')], dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2, subset=[7, 0, 2]), 20000),
    ('mark4', [S('sample.m4')], dict(sample_rate=HZ(32e6), ntrack=64, decade=2010), 160000),
    ('mark4', [S('sample.m4')], dict(sample_rate=HZ(32e6), ntrack=64, decade=2010, subset=SL(1, 8, 3)), 160000),
    ('mark4', [S('sample_32track.m4')], dict(sample_rate=HZ(32e6), ntrack=32, decade=2010), 160000),
    ('mark4', [S('sample_32track_fanout2.m4')], dict(sample_rate=HZ(16e6), ntrack=32, decade=2010), 80000),
    ('mark4', [S('sample_16track.m4')], dict(sample_rate=HZ(32e6), ntrack=16, decade=2010), 160000),
    ('mark4', [S('sample_64track_fanout2_ft.m4')], dict(sample_rate=HZ(8e6), ntrack=64, decade=2010), 40000),
    ('dada', [S('sample.dada')], {}, 16000),
    ('dada', [S('sample.dada')], dict(subset=1), 16000),
    ('vdif', [S('sample.vdif')], dict(verify=False), 40000),
    ('vdif', [S('sample.vdif')], dict(verify=True, subset=[7, 0]), 40000),
    ('mark5b', [S('sample.m5b')], dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2, verify=False), 20000),
    ('mark4', [S('sample.m4')], dict(sample_rate=HZ(32e6), ntrack=64, decade=2010, verify=False, subset=[0, 5]), 160000),
    ('dada', [S('sample_meerkat.dada')], {}, 14336),
    ('dada', [S('sample_mkbf.dada')], {}, 256),
    ('guppi', [S('sample_puppi.raw')], {}, 3840),
    ('guppi', [S('sample_puppi.raw')], dict(subset=TUP(0, [3, 1])), 3840),
)


def walk(k, fmt, args, kw, length, nops=24):
    steps, x = [open_('f', fmt, *args, 'rs', **kw), get('f.shape'), get('f.sample_shape')], 1000 + k
    for _ in range(nops):
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        offset = (x >> 24) % length
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        count = 1 + (x >> 28) % max(1, min(length - offset, length // 3))
        steps += [do('f.seek', offset), call(None, 'f.read', count), call(None, 'f.tell')]
    steps += [do('f.seek', 0), call(None, 'f.read'), close('f')]
    return steps


TS_RAW, RAW = S('gsb/sample_gsb_rawdump.timestamp'), S('gsb/sample_gsb_rawdump.dat')
TS_PH = S('gsb/sample_gsb_phased.timestamp')
PH = [[S('gsb/sample_gsb_phased.Pol-L1.dat'), S('gsb/sample_gsb_phased.Pol-L2.dat')],
      [S('gsb/sample_gsb_phased.Pol-R1.dat'), S('gsb/sample_gsb_phased.Pol-R2.dat')]]
RATE_RAW = (1e8 / 3) / 2 ** 23 * 2 ** 12 * 2          # 4 bits, 4096-byte payloads
RATE_PH = (1e8 / 3) / 2 ** 23 * 2 ** 12 / 512         # 8 bits complex, 512 channels: 8 samples per frame
GSB_WALKS = (
    ('gsb', [TS_RAW], dict(raw=RAW, sample_rate=HZ(RATE_RAW), samples_per_frame=8192), 81920),
    ('gsb', [TS_PH], dict(raw=PH, sample_rate=HZ(RATE_PH), samples_per_frame=8), 80),
    ('gsb', [TS_PH], dict(raw=PH, sample_rate=HZ(RATE_PH), samples_per_frame=8, subset=TUP(1, SL(10, 400, 7))), 80),
    ('gsb', [TS_PH], dict(raw=PH[0], sample_rate=HZ(RATE_PH), samples_per_frame=8, squeeze=False), 80),
)

CASES = [
    case('walks_over_the_samples',
         'twenty-five readers -- every sample recording, whole and with subsets of threads, channels, '
         'polarisations -- each taken through twenty-four seeks to a random place and reads of a random '
         'length, then one read of everything (the stream reader tests of every format, positions widened)',
         [walk(k, *w) for k, w in enumerate(WALKS)]),

    case('walks_over_the_gsb_samples',
         'the same for the GSB samples: the 4-bit rawdump, the phased array with both polarisations, with a '
         'polarisation and every seventh channel, with one polarisation only (gsb/tests/test_gsb.py stream reader tests)',
         [walk(50 + k, *w) for k, w in enumerate(GSB_WALKS)]),
]
