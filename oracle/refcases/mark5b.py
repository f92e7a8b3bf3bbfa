"""Mark 5B: headers, payloads, frames, binary and stream readers / writers."""
from ._dsl import *    # noqa: F401,F403

STREAM_FACTS = ('sample_rate', 'samples_per_frame', 'sample_shape', 'shape', 'size', 'ndim', 'bps', 'complex_data',
                'start_time', 'stop_time', 'time', 'fill_value', 'squeeze', 'subset', 'verify')
M5 = S('sample.m5b')
OPEN = dict(sample_rate=HZ(32e6), kday=56000, nchan=8, bps=2)
LEVELS2 = [-3.316505, -1.0, 1.0, 3.316505]

CASES = [
    case('stream_reader_facts',
         'the sample as a stream: sizes, times, reading across the frame boundary, seeking by time, the '
         'end of the file (mark5b/tests/test_mark5b.py, test_filestreamer)',
         open_('fh', 'mark5b', M5, 'rs', **OPEN),
         gets('fh', *STREAM_FACTS), get('fh.header0'), get('fh._last_header'), get('fh.dtype'),
         call('rec', 'fh.read', 12), call(None, 'fh.tell'), get('fh.time', as_='t12'), fn(None, 'as_int', V('rec')),
         do('fh.seek', 0), do('fh.seek', V('t12')), call(None, 'fh.tell'),
         do('fh.seek', 4990), call(None, 'fh.read', 20), call(None, 'fh.tell', unit='time'),
         do('fh.seek', -10, 2), call(None, 'fh.read'), call(None, 'fh.read', 1),
         do('fh.seek', 3, 'end'), call(None, 'fh.read', 1),
         close('fh'),
         open_('f2', 'mark5b', M5, 'rs', sample_rate=HZ(32e6), ref_time=TIME('2014-01-01T00:00:00'), nchan=8, bps=2,
               subset=[1, 5], squeeze=False),
         get('f2.start_time'), get('f2.sample_shape'), call(None, 'f2.read', 6), close('f2')),

    case('arguments_the_opener_needs',
         'without kday / ref_time the time cannot be resolved; nchan is required; bad modes; frame rate '
         'found by scanning when no sample rate is given (test_mark5b.py, test_stream_invalid etc.)',
         open_('a', 'mark5b', M5, 'rs', sample_rate=HZ(32e6), nchan=8, bps=2),
         open_('b', 'mark5b', M5, 'rs', sample_rate=HZ(32e6), kday=56000),
         open_('c', 'mark5b', M5, 'rs', kday=56000, nchan=8),
         get('c.sample_rate'), get('c.bps'), get('c.stop_time'), close('c'),
         open_('d', 'mark5b', M5, 's', kday=56000, nchan=8),
         open_('e', 'mark5b', M5, 'rs', kday=56000, nchan=8, bla=1),
         open_('f', 'mark5b', M5, 'rb', kday=56000, nchan=8, bps=2),
         get('f.info.readable', quiet=False), close('f')),

    case('binary_reader_and_search',
         'read_header / read_frame, find_header in both directions, locate_frames with check offsets, '
         'frame rate by scanning (test_mark5b.py, test_filereader / test_find_header / test_locate_frames)',
         open_('fb', 'mark5b', M5, 'rb', kday=56000, nchan=8, bps=2),
         call('h', 'fb.read_header'), call(None, 'fb.tell'), get('h.time'), get('h.seconds'), get('h.fraction'),
         get('h.jday'), get('h.kday'), item(None, 'h', 'frame_nr'), item(None, 'h', 'bcd_jday'),
         get('h.frame_nbytes'), get('h.payload_nbytes'), get('h.nbytes'),
         do('fb.seek', 0), call('fr', 'fb.read_frame'), call(None, 'fb.tell'), get('fr.shape'), get('fr.valid'),
         get('fr.data'), item(None, 'fr', TUP(SL(3, 6), 2)),
         do('fb.seek', 0), call(None, 'fb.get_frame_rate'),
         do('fb.seek', 10), call(None, 'fb.find_header'), call(None, 'fb.tell'),
         do('fb.seek', 10), call(None, 'fb.find_header', forward=False), call(None, 'fb.tell'),
         do('fb.seek', -10, 2), call(None, 'fb.find_header', forward=False), call(None, 'fb.tell'),
         do('fb.seek', -10, 2), call(None, 'fb.find_header', forward=True),
         do('fb.seek', 5000), call(None, 'fb.locate_frames'), call(None, 'fb.locate_frames', forward=False),
         do('fb.seek', 5000), call(None, 'fb.locate_frames', maximum=1000),
         do('fb.seek', 0), call(None, 'fb.locate_frames', check=[-1, 1, 2]),
         close('fb')),

    case('search_in_damaged_copies',
         'copies of the sample with junk in front, a cut first frame, and a zeroed sync word: where the '
         'first and last headers are found (test_mark5b.py, test_find_header with corrupted files)',
         fn('all', 'file_bytes', M5, quiet=True),
         fn(None, 'write_file', T('junk.m5b'), [HEX('ab' * 333), V('all')]),
         open_('f1', 'mark5b', T('junk.m5b'), 'rb', kday=56000, nchan=8, bps=2),
         call(None, 'f1.find_header'), call(None, 'f1.tell'), call(None, 'f1.locate_frames'),
         do('f1.seek', 0, 2), call(None, 'f1.find_header', forward=False), call(None, 'f1.tell'), close('f1'),
         open_('s1', 'mark5b', T('junk.m5b'), 'rs', **OPEN), get('s1.shape'), get('s1.start_time'),
         call(None, 's1.read', 4), close('s1'),
         fn('cut', 'file_bytes', M5, 4000, quiet=True), fn(None, 'write_file', T('cut.m5b'), [V('cut')]),
         open_('f2', 'mark5b', T('cut.m5b'), 'rb', kday=56000, nchan=8, bps=2),
         call(None, 'f2.find_header'), call(None, 'f2.tell'), close('f2'),
         fn(None, 'write_file', T('nosync.m5b'), [V('all')]), fn(None, 'patch_file', T('nosync.m5b'), 10016, HEX('00000000')),
         open_('f3', 'mark5b', T('nosync.m5b'), 'rb', kday=56000, nchan=8, bps=2),
         do('f3.seek', 10016), call(None, 'f3.find_header'), call(None, 'f3.tell'),
         do('f3.seek', 10016 + 5000), call(None, 'f3.find_header', forward=False), call(None, 'f3.tell'), close('f3')),

    case('header_times',
         'kday from a reference time, BCD day / seconds / fraction from a time and back, the 0.1 ms '
         'granularity of the stored fraction (test_mark5b.py, TestMark5B.test_header / test_header_times)',
         call('h', 'mark5b.Mark5BHeader.fromvalues', time=TIME('2014-06-13T05:30:01.000781250'), frame_rate=HZ(6400.0), user=3,
              internal_tvg=False),
         get('h'), get('h.time'), get('h.kday'), get('h.jday'), get('h.seconds'), get('h.fraction'),
         call(None, 'h.get_time', frame_rate=HZ(6400.0)),
         call('g', 'mark5b.Mark5BHeader.fromvalues', kday=56000, jday=821, seconds=19801, fraction=0.0012, frame_nr=0),
         get('g'), get('g.time'),
         [[call('k', 'mark5b.Mark5BHeader', V('h.words'), ref_time=TIME(ref)), get('k.kday'), get('k.time')]
          for ref in ('2014-01-01T00:00:00', '2016-09-01T00:00:00', '2011-06-01T00:00:00')],
         call('m', 'h.copy'), set_('m.time', TIME('2020-02-29T23:59:59.9999')), call('m', 'h.copy', quiet=True),
         call(None, 'm.set_time', TIME('2020-03-01T00:00:00.00015625'), frame_rate=HZ(6400.0)), get('m'),
         call(None, 'm.get_time', frame_rate=HZ(6400.0)),
         set_('h.time', TIME('2020-02-29T23:59:59.9999')),
         gpu=False),

    case('payload_and_frame_from_data',
         'payloads of 2-bit 8-channel and 1-bit 16-channel data and frames around them: words, bytes '
         'written, validity through the fill pattern (test_mark5b.py, TestMark5B.test_payload / test_frame)',
         let('d2', RNG(11, (5000, 8), LEVELS2)),
         call('p2', 'mark5b.Mark5BPayload.fromdata', V('d2'), bps=2), get('p2'), get('p2.shape'), get('p2.data'),
         item(None, 'p2', TUP(SL(10, 14), SL(2, 4))),
         let('d1', RNG(12, (5000, 16), [-1.0, 1.0])),
         call('p1', 'mark5b.Mark5BPayload.fromdata', V('d1'), bps=1), get('p1'), get('p1.data'),
         call('h', 'mark5b.Mark5BHeader.fromvalues', time=TIME('2014-06-13T05:30:01'), frame_nr=0),
         call('fr', 'mark5b.Mark5BFrame.fromdata', V('d2'), V('h'), bps=2), get('fr'), get('fr.valid'),
         file_('out', T('f.m5b'), 'w+b'), do('fr.tofile', V('out')),
         call('bad', 'mark5b.Mark5BFrame.fromdata', V('d2'), V('h'), bps=2, valid=False), get('bad.valid'),
         do('bad.tofile', V('out')), close('out'), digest(T('f.m5b')),
         open_('fb', 'mark5b', T('f.m5b'), 'rb', kday=56000, nchan=8, bps=2),
         call('r1', 'fb.read_frame'), eq(V('r1'), V('fr')), call('r2', 'fb.read_frame'), get('r2.valid'),
         get('r2.data'), close('fb'),
         set_('fr.valid', False), get('fr.valid'), get('fr.payload'), get('fr.data')),

    case('stream_writer_and_incomplete_frame',
         'a writer fed a frame and a bit pads the second frame and marks it invalid; the file written '
         'from the sample equals the sample (test_mark5b.py, test_stream_writer / test_incomplete_stream)',
         open_('fr', 'mark5b', M5, 'rs', **OPEN), call('all', 'fr.read'),
         open_('fw', 'mark5b', T('copy.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         gets('fw', 'sample_rate', 'samples_per_frame', 'sample_shape', 'start_time'),
         do('fw.write', V('all')), get('fw.time'), close('fw'), digest(T('copy.m5b')),
         item('part', 'all', SL(None, 5010), quiet=True),
         open_('fp', 'mark5b', T('part.m5b'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6), nchan=8, bps=2),
         do('fp.write', V('part')), close('fp'), digest(T('part.m5b')),
         [[open_('f', 'mark5b', T('part.m5b'), 'rs', fill_value=fv, **OPEN), get('f.shape'),
           call(None, 'f.read', 5000), call('tail', 'f.read'), fn(None, 'allclose_to', V('tail'), fv), close('f')]
          for fv in (0.0, -999.0)],
         open_('fk', 'mark5b', T('kw.m5b'), 'ws', sample_rate=HZ(32e6), nchan=8, bps=2,
               time=TIME('2014-06-13T05:30:01')),
         do('fk.write', V('all')), close('fk'), digest(T('kw.m5b')),
         close('fr')),

    case('pickle_and_sequence',
         'a pickled reader continues where the original stood; two files read as one stream '
         '(test_mark5b.py, test_pickle; baseband/tests/test_sequential_baseband.py)',
         open_('fh', 'mark5b', M5, 'rs', **OPEN), do('fh.seek', 4321),
         fn('fp', 'pickle_roundtrip', V('fh'), quiet=True), call(None, 'fp.tell'), call(None, 'fp.read', 7),
         close('fp'), do('fh.seek', 0), call('all', 'fh.read'),
         open_('fw', 'mark5b', T('p{file_nr:d}.m5b'), 'ws', header0=V('fh.header0'), sample_rate=HZ(32e6), nchan=8,
               bps=2, file_size=2 * 10016),
         do('fw.write', V('all')), close('fw'), listdir(), close('fh'),
         open_('fs', 'mark5b', [T('p0.m5b'), T('p1.m5b')], 'rs', **OPEN), get('fs.shape'),
         do('fs.seek', 9995), call(None, 'fs.read', 10), close('fs')),
]
