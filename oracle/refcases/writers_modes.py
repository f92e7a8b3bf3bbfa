"""Every encoder mode through its stream writer: bytes on disk and the samples read back."""
from ._dsl import *    # noqa: F401,F403

L2 = [-3.316505, -1.0, 1.0, 3.316505]
L1 = [-1.0, 1.0]
L4 = [-8.0, -5.0, -1.0, 0.0, 2.0, 7.0]
L8 = [-128.0, -35.0, -3.0, 0.0, 8.0, 127.0]

CASES = [
    case('mark4_all_track_layouts',
         'the five Mark 4 layouts the reference encodes: 16 tracks (2 ch), 32 tracks fan-out 4 (4 ch) and 2 '
         '(8 ch), 64 tracks fan-out 4 (8 ch), and the 64-track fan-out 2 "ft" layout (16 ch, header taken '
         'from its sample): two frames each (mark4/tests/test_mark4.py, the per-layout writer tests)',
         [[let('d', RNG(seed, (2 * spf, nchan), L2)),
           open_('fw', 'mark4', T(name), 'ws', sample_rate=HZ(rate), ntrack=ntrack, fanout=fanout, nchan=nchan, bps=2,
                 time=TIME('2015-03-04T05:06:07.0000')),
           get('fw.header0'), get('fw.samples_per_frame'), do('fw.write', V('d')), close('fw'), digest(T(name)),
           open_('fr', 'mark4', T(name), 'rs', ntrack=ntrack, decade=2010, sample_rate=HZ(rate)), get('fr.shape'),
           call('b', 'fr.read'), item(None, 'b', SL(160 * fanout - 2, 160 * fanout + 3)), close('fr')]
          for seed, name, ntrack, fanout, nchan, spf, rate in (
              (1, 't16.m4', 16, 4, 2, 80000, 32e6), (2, 't32f4.m4', 32, 4, 4, 80000, 32e6),
              (3, 't32f2.m4', 32, 2, 8, 40000, 16e6), (4, 't64f4.m4', 64, 4, 8, 80000, 32e6))],
         open_('fs', 'mark4', S('sample_64track_fanout2_ft.m4'), 'rs', sample_rate=HZ(8e6), decade=2010),
         get('fs.shape'), call('dft', 'fs.read', 40000), get('fs.header0', as_='hft'),
         open_('fw', 'mark4', T('ft.m4'), 'ws', header0=V('hft'), sample_rate=HZ(8e6)), do('fw.write', V('dft')),
         close('fw'), digest(T('ft.m4')),
         fn('orig', 'file_bytes', S('sample_64track_fanout2_ft.m4'), quiet=True),
         open_('fb', 'mark4', S('sample_64track_fanout2_ft.m4'), 'rb', ntrack=64, decade=2010),
         call('offs', 'fb.locate_frames'), close('fb'), close('fs')),

    case('mark5b_widths_and_channels',
         'Mark 5B writers: 1-bit 16 channels, 2-bit 1 / 2 / 4 channels (sign / magnitude bit order, the '
         'payload always 10000 bytes) (mark5b/tests/test_mark5b.py, writer tests by nchan / bps)',
         [[let('d', RNG(seed, (2 * 80000 // (bps * nchan), nchan), L1 if bps == 1 else L2)),
           open_('fw', 'mark5b', T(name), 'ws', sample_rate=HZ(1e6), nchan=nchan, bps=bps, time=TIME('2018-05-06T07:08:09'),
                 squeeze=False),
           get('fw.samples_per_frame'), do('fw.write', V('d')), close('fw'), digest(T(name)),
           open_('fr', 'mark5b', T(name), 'rs', sample_rate=HZ(1e6), kday=58000, nchan=nchan, bps=bps, squeeze=False),
           get('fr.shape'), call('b', 'fr.read'), eq(V('b'), V('d')), close('fr')]
          for seed, name, nchan, bps in ((5, 'b1c16.m5b', 16, 1), (6, 'b2c1.m5b', 1, 2), (7, 'b2c2.m5b', 2, 2),
                                         (8, 'b2c4.m5b', 4, 2))]),

    case('gsb_dada_guppi_shapes',
         'GSB rawdump (4-bit real) and phased with one file per polarisation (8-bit complex); DADA real '
         '8-bit one polarisation; GUPPI 8 channels 2 polarisations time-first: digests and read-back '
         '(gsb/tests/test_gsb.py, dada/tests/test_dada.py, guppi/tests/test_guppi.py writer tests)',
         let('dr', RNG(9, (2 * 8192, 1), L4)),
         open_('fw', 'gsb', T('r.timestamp'), 'ws', raw=T('r.dat'), sample_rate=HZ((1e8 / 3) / 2 ** 23 * 8192),
               samples_per_frame=8192, nchan=1, bps=4, complex_data=False, time=TIME('2015-04-27T13:15:00'), squeeze=False),
         get('fw.header0'), do('fw.write', V('dr')), close('fw'), digest(T('r.dat')), digest(T('r.timestamp')),
         let('dp', RNG(10, (16, 2, 512), L8, complex=True)),
         open_('fw', 'gsb', T('p.timestamp'), 'ws', raw=[[T('pL.dat')], [T('pR.dat')]],
               sample_rate=HZ((1e8 / 3) / 2 ** 23 * 8), samples_per_frame=8, nchan=512, bps=8, complex_data=True,
               time=TIME('2013-07-27T21:23:55.3241088')),
         get('fw.header0'), get('fw.sample_shape'), do('fw.write', V('dp')), close('fw'), digest(T('pL.dat')),
         digest(T('pR.dat')), digest(T('p.timestamp')),
         let('dd', RNG(11, (3000, 1, 1), L8)),
         open_('fw', 'dada', T('real.dada'), 'ws', time=TIME('2013-07-02T01:39:20'), sample_rate=HZ(16e6),
               samples_per_frame=1500, npol=1, nchan=1, bps=8, complex_data=False, squeeze=False),
         get('fw.header0'), do('fw.write', V('dd')), close('fw'), digest(T('real.dada')),
         open_('fr', 'dada', T('real.dada'), 'rs', squeeze=False), get('fr.shape'), get('fr.dtype'), call('b', 'fr.read'),
         eq(V('b'), V('dd')), close('fr')),
]
