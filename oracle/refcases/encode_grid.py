"""Encoders and decoders over a grid of sample widths and shapes, on values that sit on and next to the
steps of each encoder: what is written for them (file digests) and what is read back."""
from ._dsl import *    # noqa: F401,F403

# values on, just below and just above the decision steps of the 2-, 4- and 8-bit encoders
TWO = [-3.316505, -2.1, -2.0, -1.99999, -1.0, -0.00001, 0.0, 0.00001, 1.0, 1.99999, 2.0, 2.1, 3.316505, 40.0, -40.0]
FOUR = [-9.0, -8.5, -8.0, -7.5, -7.49999, -2.5, -1.5, -0.5, -0.49999, 0.0, 0.49999, 0.5, 1.5, 2.5, 6.5, 7.0, 7.49999, 7.5, 8.0, 30.0]
EIGHT = [-300.0, -128.5, -128.0, -127.5, -2.5, -1.5, -0.5, -0.49999, 0.0, 0.49999, 0.5, 1.5, 2.5, 126.5, 127.0, 127.49999, 127.5,
         128.0, 300.0]
ONE = [-5.0, -1.0, -0.00001, 0.0, 0.00001, 1.0, 5.0]
BY_BPS = {1: ONE, 2: TWO, 4: FOUR, 8: EIGHT}

VDIF_GRID = [(bps, cplx, nchan, nthread, edv)
             for bps in (1, 2, 4, 8) for cplx in (False, True)
             for nchan, nthread, edv in ((1, 1, 0), (2, 4, 1), (16, 2, 3) if bps == 2 else (8, 1, False), (4, 8, 0))]


def vdif_steps(k, bps, cplx, nchan, nthread, edv):
    ncomp = 2 if cplx else 1
    if edv == 3:                    # EDV 3 frames hold 1000 or 5000 bytes
        spf = 5000 * 8 // bps // nchan // ncomp
    else:                           # 64 bytes of payload
        spf = 64 * 8 // bps // nchan // ncomp
    rate = spf * 1000.              # (whole kHz: EDV 1 and 3 store the rate)
    name = 'g%d.vdif' % k
    return [let('d', RNG(500 + k, (2 * spf, nthread, nchan), BY_BPS[bps], complex=cplx)),
            open_('fw', 'vdif', T(name), 'ws', sample_rate=HZ(rate), nthread=nthread, nchan=nchan, bps=bps,
                  complex_data=cplx, edv=edv, station='gx', samples_per_frame=spf, time=TIME('2019-07-01T00:00:00'),
                  squeeze=False),
            do('fw.write', V('d')), close('fw'), digest(T(name)),
            open_('fr', 'vdif', T(name), 'rs', squeeze=False, **({} if edv in (1, 3) else dict(sample_rate=HZ(rate)))),
            get('fr.shape'), call(None, 'fr.read'), close('fr')]


CASES = [
    case('vdif_widths_shapes_and_steps',
         'VDIF streams of 1, 2, 4 and 8 bits, real and complex, 1 to 16 channels, 1 to 8 threads, legacy and EDV '
         '0 / 1 / 3 headers, written from values on and next to the encoders\' steps and read back '
         '(vdif/tests/test_vdif.py, the encoder / decoder round trips by bps; payload.py encode_*)',
         [vdif_steps(k, *cfg) for k, cfg in enumerate(VDIF_GRID)]),

    case('mark5b_and_mark4_steps',
         'the 1- and 2-bit encoders of Mark 5B (sign / magnitude order) and the 2-bit one of Mark 4 (tracks by '
         'fan-out, the four layouts the reference encodes from keywords) on the same values (mark5b/tests/test_mark5b.py and '
         'mark4/tests/test_mark4.py, encoder tests)',
         [[let('d', RNG(600 + k, (80000 // bps // nchan * 2, nchan), BY_BPS[bps])),
           open_('fw', 'mark5b', T('g%d.m5b' % k), 'ws', sample_rate=HZ(32e6), nchan=nchan, bps=bps,
                 time=TIME('2014-06-13T05:30:01'), squeeze=False),
           do('fw.write', V('d')), close('fw'), digest(T('g%d.m5b' % k)),
           open_('fr', 'mark5b', T('g%d.m5b' % k), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=nchan, bps=bps,
                 squeeze=False),
           call(None, 'fr.read'), close('fr')]
          for k, (bps, nchan) in enumerate(((1, 1), (1, 2), (1, 32), (2, 1), (2, 2), (2, 16)))],
         [[let('d', RNG(650 + k, (2 * 20000 * fanout, nchan), TWO)),
           open_('fw', 'mark4', T('g%d.m4' % k), 'ws', sample_rate=HZ(32e6), ntrack=ntrack, fanout=fanout,
                 nchan=nchan, bps=2, time=TIME('2014-06-16T07:38:12.47500'), squeeze=False),
           do('fw.write', V('d')), close('fw'), digest(T('g%d.m4' % k)),
           open_('fr', 'mark4', T('g%d.m4' % k), 'rs', sample_rate=HZ(32e6), ntrack=ntrack, decade=2010, squeeze=False),
           call(None, 'fr.read'), close('fr')]
          for k, (ntrack, fanout, nchan) in enumerate(((64, 4, 8), (32, 4, 4), (32, 2, 8), (16, 4, 2)))]),

    case('byte_format_steps',
         'the 8-bit encoders of DADA and GUPPI on the same values: rounding and clipping, real and complex, one '
         'and two polarisations, channels (dada / guppi payload tests)',
         [[let('d', RNG(700 + k, (256, npol, nchan), EIGHT, complex=cplx)),
           open_('fw', 'dada', T('g%d.dada' % k), 'ws', time=TIME('2013-07-02T01:39:20'), sample_rate=HZ(16e6),
                 samples_per_frame=128, npol=npol, nchan=nchan, bps=8, complex_data=cplx, squeeze=False),
           do('fw.write', V('d')), close('fw'), digest(T('g%d.dada' % k)),
           open_('fr', 'dada', T('g%d.dada' % k), 'rs', squeeze=False), call(None, 'fr.read'), close('fr')]
          for k, (npol, nchan, cplx) in enumerate(((1, 1, False), (2, 1, True), (2, 4, True), (1, 4, False)))],
         open_('fg', 'guppi', S('sample_puppi.raw'), 'rs'), call('hg', 'fg.header0.copy'), close('fg'),
         set_('hg.overlap', 0),
         [[set_('hg.nchan', nchan), set_('hg.samples_per_frame', 128),
           let('d', RNG(750 + k, (256, 2, nchan), EIGHT, complex=True)),
           open_('fw', 'guppi', T('g%d.raw' % k), 'ws', header0=V('hg'), squeeze=False),
           do('fw.write', V('d')), close('fw'), digest(T('g%d.raw' % k)),
           open_('fr', 'guppi', T('g%d.raw' % k), 'rs', squeeze=False), call(None, 'fr.read'), close('fr')]
          for k, nchan in enumerate((2, 4, 8))]),
]
