"""Odds and ends of the readers: scanning the less regular samples, out= of every kind, holes in
file sequences, what writers refuse."""
from ._dsl import *    # noqa: F401,F403

CASES = [
    case('scanning_irregular_vdif',
         'thread ids, frame rates and frame sets of the samples whose layout is not the usual one: MWA '
         '(one thread, 8-bit complex), ARO CHIME (2 threads x 1024 channels, one sample per frame), VLBA '
         '(threads out of step), 1-bit (vdif/tests/test_vdif.py, get_thread_ids / get_frame_rate / '
         'read_frameset on the other samples)',
         [[open_('fb', 'vdif', S(name), 'rb'), call(None, 'fb.get_thread_ids'), call(None, 'fb.get_frame_rate'),
           do('fb.seek', 0), call('fs', 'fb.read_frameset'), call(None, 'fb.tell'), get('fs.shape'),
           item(None, 'fs', 'thread_id'), item(None, 'fs', 'frame_nr'), get('fs.valid'), get('fs.data'),
           call('fs2', 'fb.read_frameset'), item(None, 'fs2', 'frame_nr'), call(None, 'fb.tell'),
           get('fb.info.number_of_frames'), get('fb.info.number_of_framesets'), close('fb')]
          for name in ('sample_mwa.vdif', 'sample_arochime.vdif', 'sample_vlbi.vdif', 'sample_bps1.vdif')]),

    case('read_into_buffers_of_the_caller',
         'read(out=...) for real and complex streams, squeezed and not, subsets; wrong shapes and a buffer '
         'longer than what is left (base/tests/test_base.py and the formats\' out= cases)',
         open_('fd', 'dada', S('sample.dada'), 'rs'),
         let('o1', ZEROS((100, 2), 'c8')), do('fd.read', out=V('o1')), get('o1'), call(None, 'fd.tell'),
         let('o2', ZEROS((100, 2, 1), 'c8')), do('fd.read', out=V('o2')),
         let('o3', ZEROS((100, 2), 'f4')), do('fd.seek', 0), do('fd.read', out=V('o3')), get('o3'),
         do('fd.seek', -50, 2), let('o4', ZEROS((100, 2), 'c8')), do('fd.read', out=V('o4')), call(None, 'fd.tell'),
         close('fd'),
         open_('fg', 'guppi', S('sample_puppi.raw'), 'rs', subset=TUP(0, SL(1, 3))),
         let('g1', ZEROS((1000, 2), 'c8')), do('fg.seek', 500), do('fg.read', out=V('g1')), get('g1'), close('fg'),
         open_('fm', 'mark4', S('sample.m4'), 'rs', ntrack=64, decade=2010, squeeze=False, subset=[0, 7]),
         let('m1', ZEROS((700, 2), 'f4')), do('fm.read', out=V('m1')), get('m1'), close('fm'),
         open_('f5', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         let('b1', ZEROS((4999, 8), 'f8')), do('f5.read', out=V('b1')), get('b1'), close('f5')),

    case('holes_in_sequences',
         'a template whose second file is missing stops at the first; a list that skips a file reads what '
         'it is given; the reader of a sequence seeks across files (helpers/tests and vdif template tests)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', T('h{file_nr:d}.vdif'), 'ws', header0=V('fr.header0'), nthread=8, file_size=4 * 5032),
         do('fw.write', V('d')), close('fw'), listdir(), close('fr'),
         fn('gone', 'file_bytes', T('h1.vdif'), quiet=True), fn(None, 'truncate', T('h1.vdif'), 0),
         open_('f1', 'vdif', T('h{file_nr:d}.vdif'), 'rs'), get('f1.shape'), call(None, 'f1.read', 5), close('f1'),
         fn(None, 'write_file', T('h1.vdif'), [V('gone')]),
         open_('f2', 'vdif', [T('h0.vdif'), T('h1.vdif'), T('h2.vdif'), T('h3.vdif')], 'rs'), get('f2.shape'),
         do('f2.seek', 9990), call(None, 'f2.read', 20), do('f2.seek', -5, 2), call(None, 'f2.read'), close('f2'),
         open_('f3', 'vdif', [T('h0.vdif'), T('h2.vdif')], 'rs', verify=True), get('f3.shape'), call(None, 'f3.read', 12000),
         close('f3')),

    case('what_writers_refuse',
         'data of the wrong trailing shape, writing after close, a header that cannot hold the samples '
         'per frame asked for, complex data into a real stream (stream writer argument checks of every format)',
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read', 20000),
         open_('f5', 'mark5b', S('sample.m5b'), 'rs', sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2),
         get('f5.header0', 'hm', quiet=True), close('f5'),
         open_('fw', 'vdif', T('w.vdif'), 'ws', header0=V('fr.header0'), nthread=8),
         item('bad', 'd', TUP(SL(None), SL(0, 7)), quiet=True), do('fw.write', V('bad')),
         do('fw.write', V('d')), close('fw'), do('fw.write', V('d')), get('fw.closed'),
         open_('fx', 'vdif', T('x.vdif'), 'ws', header0=V('fr.header0'), nthread=8, bla=3),
         open_('fy', 'vdif', T('y.vdif'), 'ws', nthread=8, samples_per_frame=20001, nchan=1, bps=2, complex_data=False,
               edv=3, station=1, time=TIME('2014-06-16T05:56:07'), sample_rate=HZ(32e6)),
         open_('fz', 'mark5b', T('z.m5b'), 'ws', sample_rate=HZ(32e6), nchan=8, bps=3, time=TIME('2014-06-13T05:30:01')),
         open_('fz4', 'mark5b', T('z4.m5b'), 'ws', sample_rate=HZ(32e6), nchan=8, bps=4, time=TIME('2014-06-13T05:30:01')),
         open_('fz5', 'mark5b', T('z5.m5b'), 'ws', header0=V('hm'), sample_rate=HZ(32e6), nchan=8, bla=1),
         open_('fy2', 'vdif', T('y2.vdif'), 'ws', nthread=1, samples_per_frame=20032, nchan=1, bps=2, complex_data=False,
               edv=3, station=1, time=TIME('2014-06-16T05:56:07'), sample_rate=HZ(32e6)),
         open_('fy3', 'vdif', T('y3.vdif'), 'ws', nthread=1, samples_per_frame=20032, nchan=1, bps=2, complex_data=False,
               edv=1, station=1, time=TIME('2014-06-16T05:56:07'), sample_rate=HZ(32e6)),
         get('fy3.header0'), close('fy3'),
         open_('fq', 'mark4', T('q.m4'), 'ws', sample_rate=HZ(32e6), ntrack=48, fanout=4, nchan=8, bps=2,
               time=TIME('2014-06-16T07:38:12.4750')),
         # (the reference has coders for five Mark 4 layouts; this package packs by bit maps and takes any)
         open_('fq2', 'mark4', T('q2.m4'), 'ws', sample_rate=HZ(32e6), ntrack=64, fanout=2, nchan=16, bps=2,
               time=TIME('2014-06-16T07:38:12.4750'), we_may_manage=True, quiet=True),
         call('h16', 'vdif.VDIFHeader.fromvalues', edv=0, bps=16, nchan=2, complex_data=True, samples_per_frame=48,
              station='aa', time=TIME('2015-01-01T00:00:00'), frame_rate=HZ(100.)),
         call(None, 'vdif.VDIFPayload.fromdata', RNG(5, (48, 2), [-3.0, 0.0, 2.0], complex=True), V('h16')),
         close('fr')),
]
