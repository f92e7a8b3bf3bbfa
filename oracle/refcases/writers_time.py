"""Stream writers across second, day and year boundaries and block counters: header fields that tick."""
from ._dsl import *    # noqa: F401,F403

L2 = [-3.316505, -1.0, 1.0, 3.316505]
L8 = [-120.0, -35.0, -3.6, -0.5, 0.0, 0.7, 3.3, 50.0, 126.0]

CASES = [
    case('vdif_seconds_and_epochs',
         'VDIF frames count seconds from a half-year epoch and frames within the second: a stream that '
         'crosses a second, one that crosses New Year (epoch kept from header0), and the Mark 5B flavour '
         '(EDV 0xab) -- digests and the first / last header read back (vdif/tests/test_vdif.py, stream writer '
         'time handling; baseband/tests/test_conversion.py, TestVDIFMark5B)',
         [[let('d', RNG(seed, (nframes * spf, nthread, nchan), L2)),
           open_('fw', 'vdif', T(name), 'ws', sample_rate=HZ(rate), nthread=nthread, nchan=nchan, bps=2,
                 complex_data=False, edv=edv, station=stn, samples_per_frame=spf, time=TIME(t0), squeeze=False),
           get('fw.header0'), do('fw.write', V('d')), get('fw.time'), close('fw'), digest(T(name)),
           open_('fb', 'vdif', T(name), 'rb'), call('h0', 'fb.read_header'), get('h0.time'),
           item(None, 'h0', 'seconds'), item(None, 'h0', 'frame_nr'), item(None, 'h0', 'ref_epoch'),
           do('fb.seek', -fnb, 2), call('hl', 'fb.read_header'), get('hl.time'), item(None, 'hl', 'seconds'),
           item(None, 'hl', 'frame_nr'), item(None, 'hl', 'thread_id'), close('fb')]
          for seed, name, nframes, spf, nthread, nchan, edv, stn, rate, t0, fnb in (
              (1, 'sec.vdif', 6, 1000, 2, 4, 1, 'xy', 4000., '2015-05-31T23:59:59.250', 1032),
              (2, 'year.vdif', 4, 4000, 1, 1, 3, 65532, 8000., '2019-12-31T23:59:59.000', 1032),
              (3, 'epoch.vdif', 4, 320, 2, 2, 0, 'aa', 640., '2016-06-30T23:59:59.000', 192))],
         open_('fm', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         call('dm', 'fm.read'),
         call('hab', 'vdif.VDIFHeader.from_mark5b_header', V('fm.header0'), nchan=8, bps=2),
         open_('fw', 'vdif', T('ab.vdif'), 'ws', header0=V('hab'), nthread=1, sample_rate=HZ(32e6)),
         do('fw.write', V('dm')), close('fw'), digest(T('ab.vdif')),
         open_('fb', 'vdif', T('ab.vdif'), 'rb'), call('f0', 'fb.read_frame'), get('f0.header'), get('f0.header.edv'),
         get('f0.data'), do('fb.seek', -10032, 2), call('fl', 'fb.read_frame'), get('fl.header.time'), get('fl.data'),
         close('fb'), close('fm')),

    case('mark5b_and_mark4_clocks',
         'Mark 5B: BCD day, second and fraction plus a frame counter that restarts every second, across a '
         'day boundary, with the user word; Mark 4: BCD time with millisecond digits across a minute and a '
         'year, 16 and 64 tracks (mark5b/tests/test_mark5b.py and mark4/tests/test_mark4.py, stream writer tests)',
         let('d5', RNG(4, (6 * 5000, 8), L2)),
         open_('fw', 'mark5b', T('day.m5b'), 'ws', sample_rate=HZ(10000.), nchan=8, bps=2,
               time=TIME('2017-12-31T23:59:58.5'), user=0xbead, internal_tvg=True),
         get('fw.header0'), do('fw.write', V('d5')), get('fw.time'), close('fw'), digest(T('day.m5b')),
         open_('fb', 'mark5b', T('day.m5b'), 'rb', ref_time=TIME('2017-12-01T00:00:00'), nchan=8, bps=2),
         repeat(6, call('h', 'fb.read_header'), get('h.time'), item(None, 'h', 'frame_nr'), get('h.jday'),
                do('fb.seek', 10000, 1)),
         close('fb'),
         [[let('d4', RNG(seed, (nfr * spf, nchan), L2)),
           open_('fw', 'mark4', T(name), 'ws', sample_rate=HZ(rate), ntrack=ntrack, fanout=fanout, nchan=nchan, bps=2,
                 time=TIME(t0)),
           get('fw.header0'), do('fw.write', V('d4')), get('fw.time'), close('fw'), digest(T(name)),
           open_('fr', 'mark4', T(name), 'rs', ntrack=ntrack, decade=dec), get('fr.start_time'), get('fr.stop_time'),
           get('fr._last_header'), get('fr.shape'), close('fr')]
          for seed, name, nfr, spf, ntrack, fanout, nchan, rate, t0, dec in (
              (5, 'min.m4', 5, 80000, 64, 4, 8, 32e6, '2014-06-16T07:38:59.9950', 2010),
              (6, 'year.m4', 4, 80000, 16, 4, 2, 32e6, '2019-12-31T23:59:59.9950', 2010))]),

    case('block_counters',
         'DADA files count bytes from the start of the observation (OBS_OFFSET) and are numbered '
         '(dada/tests/test_dada.py, multi-file writer tests)',
         let('dd', RNG(7, (4 * 500, 2), L8, complex=True)),
         open_('fw', 'dada', T('o{frame_nr:02d}.dada'), 'ws', time=TIME('2013-07-02T01:39:20'), sample_rate=HZ(1000.),
               samples_per_frame=500, npol=2, nchan=1, bps=8, complex_data=True, instrument='gen'),
         do('fw.write', V('dd')), close('fw'), listdir(),
         [[digest(T('o%02d.dada' % k)), file_('f', T('o%02d.dada' % k), 'rb'),
           call('h', 'dada.DADAHeader.fromfile', V('f')), close('f'), item(None, 'h', 'OBS_OFFSET'),
           item(None, 'h', 'FILE_NUMBER'), get('h.time')] for k in range(4)]),
]
