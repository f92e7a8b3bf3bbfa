"""Mark 4: track layouts, headers inside the frames, binary and stream readers / writers."""
from ._dsl import *    # noqa: F401,F403

STREAM_FACTS = ('sample_rate', 'samples_per_frame', 'sample_shape', 'shape', 'size', 'ndim', 'bps', 'complex_data',
                'start_time', 'stop_time', 'time', 'fill_value', 'squeeze', 'subset', 'verify')
M4 = S('sample.m4')
LEVELS2 = [-3.316505, -1.0, 1.0, 3.316505]
LAYOUTS = (('sample.m4', 64, 32e6), ('sample_32track.m4', 32, 32e6), ('sample_32track_fanout2.m4', 32, 16e6),
           ('sample_16track.m4', 16, 32e6), ('sample_64track_fanout2_ft.m4', 64, 8e6))

CASES = [
    case('number_of_tracks_found',
         'determine_ntrack leaves the raw pointer at the first frame and agrees with locate_frames for a '
         'given ntrack, from the start and from inside the first frame '
         '(mark4/tests/test_mark4.py, test_determine_ntrack)',
         [[open_('fa', 'mark4', S(name), 'rb', ntrack=ntrack), do('fa.seek', start),
           call(None, 'fa.locate_frames'), close('fa'),
           open_('fn', 'mark4', S(name), 'rb'), get('fn.ntrack'), do('fn.seek', start),
           call(None, 'fn.determine_ntrack'), get('fn.ntrack'), call(None, 'fn.fh_raw.tell'), close('fn')]
          for name, ntrack, start in (('sample.m4', 64, 0), ('sample_32track.m4', 32, 10000),
                                      ('sample_32track_fanout2.m4', 32, 0), ('sample_16track.m4', 16, 0))]),

    case('every_track_layout_as_a_stream',
         'each sample recording (64 / 32 / 16 tracks, fan-out 4 and 2, the ft variant) read as a stream: '
         'shape, times, the first samples after the header gap, continuity check '
         '(test_mark4.py, Test32TrackFanout4 / Test32TrackFanout2 / Test16TrackFanout4 / Test64TrackFanout2FT)',
         [[open_('fs', 'mark4', S(name), 'rs', sample_rate=HZ(rate), decade=2010),
           gets('fs', *STREAM_FACTS), get('fs.header0'), get('fs.header0.ntrack'), get('fs.header0.fanout'),
           get('fs.header0.nchan'), get('fs.header0.time'),
           call(None, 'fs.read', 700), do('fs.seek', -20, 2), call(None, 'fs.read'), get('fs.info.readable'),
           item(None, 'fs.info.checks', 'continuous'), close('fs')]
          for name, ntrack, rate in LAYOUTS]),

    case('stream_positions_and_subsets',
         'sample.m4: frame rate found when no sample rate is given, seek by time into the second frame, '
         'channel subsets, squeeze (test_mark4.py, test_filestreamer / test_stream_reader subset)',
         open_('fh', 'mark4', M4, 'rs', ntrack=64, decade=2010),
         get('fh.sample_rate'), get('fh.shape'), get('fh._last_header'), get('fh.dtype'),
         call('rec', 'fh.read', 642), call(None, 'fh.tell'), get('fh.time', as_='t'),
         item(None, 'rec', SL(636, 642)),
         do('fh.seek', 80000 + 639), call(None, 'fh.read', 2), call(None, 'fh.tell', unit='time'),
         do('fh.seek', 0), do('fh.seek', V('t')), call(None, 'fh.tell'),
         do('fh.seek', -10, 2), call(None, 'fh.read', 11), close('fh'),
         open_('f2', 'mark4', M4, 'rs', ntrack=64, decade=2010, subset=[0, 3]), get('f2.sample_shape'),
         do('f2.seek', 640), call(None, 'f2.read', 5), close('f2'),
         open_('f3', 'mark4', M4, 'rs', ntrack=64, ref_time=TIME('2013-01-01T00:00:00'), subset=5), get('f3.sample_shape'),
         get('f3.start_time'), do('f3.seek', 640), call(None, 'f3.read', 5), close('f3'),
         open_('f4', 'mark4', M4, 'rs', ntrack=64),
         open_('f5', 'mark4', M4, 's')),

    case('binary_reader_and_headers',
         'frames at 0xa88, header fields and times with decade / ref_time, frame rate, locate_frames '
         'forward and backward (test_mark4.py, test_filereader / test_header / test_locate_frames)',
         open_('fb', 'mark4', M4, 'rb', ntrack=64, decade=2010),
         call(None, 'fb.locate_frames'), do('fb.seek', 0xa88),
         call('h', 'fb.read_header'), call(None, 'fb.tell'),
         gets('h', 'ntrack', 'fanout', 'nchan', 'bps', 'samples_per_frame', 'frame_nbytes', 'payload_nbytes', 'nbytes',
              'decade', 'time', 'fraction', 'track_id', 'converters'),
         item(None, 'h', 'bcd_unit_year'), item(None, 'h', 'bcd_day'), item(None, 'h', 'bcd_track_id'),
         item(None, 'h', 'lsb_output'), item(None, 'h', 'converter_id'), item(None, 'h', 'communication_error'),
         do('fb.seek', 0xa88), call('fr', 'fb.read_frame'), call(None, 'fb.tell'),
         get('fr.shape'), get('fr.valid'), get('fr.sample_shape'),
         item(None, 'fr', SL(636, 644)), item(None, 'fr', TUP(SL(640, 643), SL(2, 5))), item(None, 'fr', 10),
         do('fb.seek', 0), call(None, 'fb.get_frame_rate'), call(None, 'fb.find_header'), call(None, 'fb.tell'),
         do('fb.seek', 0xa88 + 100), call(None, 'fb.locate_frames'), call(None, 'fb.locate_frames', forward=False),
         do('fb.seek', -100, 2), call(None, 'fb.locate_frames', forward=False),
         do('fb.seek', -100, 2), call(None, 'fb.find_header', forward=True),
         close('fb'),
         open_('fr2', 'mark4', M4, 'rb', ntrack=64, ref_time=TIME('2018-01-01T00:00:00')), do('fr2.seek', 0xa88),
         call('h2', 'fr2.read_header'), get('h2.decade'), get('h2.time'), close('fr2'),
         open_('fr3', 'mark4', M4, 'rb', ntrack=64), do('fr3.seek', 0xa88), call('h3', 'fr3.read_header'),
         get('h3.decade'), item(None, 'h3', 'bcd_day'), close('fr3')),

    case('search_around_damage',
         'junk before the first frame, a frame cut in two, a copy with the sync pattern of the second frame '
         'broken: where frames are located (test_mark4.py, test_find_header)',
         open_('fb', 'mark4', M4, 'rb', ntrack=64, decade=2010), do('fb.seek', 0xa88),
         call('fr0', 'fb.read_frame'), call('fr1', 'fb.read_frame'), close('fb'),
         file_('o1', T('short.m4'), 'w+b'), do('o1.write', HEX('5a' * 77)), do('fr0.tofile', V('o1')),
         do('fr1.tofile', V('o1')), close('o1'),
         open_('f1', 'mark4', T('short.m4'), 'rb', ntrack=64, decade=2010),
         call(None, 'f1.locate_frames'), call(None, 'f1.find_header'), call(None, 'f1.tell'),
         do('f1.seek', 0, 2), call(None, 'f1.locate_frames', forward=False),
         do('f1.seek', 200000), call(None, 'f1.find_header', forward=False), call(None, 'f1.tell'), close('f1'),
         fn(None, 'patch_file', T('short.m4'), 77 + 160000 + 64 * 8 + 40, HEX('00' * 16)),
         open_('f2', 'mark4', T('short.m4'), 'rb', ntrack=64, decade=2010),
         call(None, 'f2.locate_frames'), do('f2.seek', 100000), call(None, 'f2.locate_frames'),
         do('f2.seek', 100000), call(None, 'f2.find_header', forward=True),
         do('f2.seek', 0, 2), call(None, 'f2.find_header', forward=False), call(None, 'f2.tell'), close('f2'),
         fn(None, 'truncate', T('short.m4'), 77 + 160000 + 64 * 20 - 8),
         open_('f3', 'mark4', T('short.m4'), 'rb', ntrack=64, decade=2010),
         do('f3.seek', 0, 2), call(None, 'f3.find_header', forward=False), call(None, 'f3.tell'), close('f3')),

    case('payload_frame_from_data',
         'payloads and frames built from data for 64 tracks fan-out 4 (8 channels) and 32 tracks fan-out 2 '
         '(8 channels): words, header gap filled, bytes written (test_mark4.py, test_payload / test_frame)',
         [[let('d', RNG(seed, (nsamp, nchan), LEVELS2)),
           call('h', 'mark4.Mark4Header.fromvalues', ntrack=ntrack, fanout=fanout, bps=2, decade=2010,
                time=TIME('2014-06-16T07:38:12.47500'), nchan=nchan),
           get('h.samples_per_frame'), get('h.time'), get('h'),
           call('fr', 'mark4.Mark4Frame.fromdata', V('d'), V('h')), get('fr.payload'), get('fr.valid'),
           item(None, 'fr', SL(160 * fanout - 2, 160 * fanout + 2)),
           file_('out', T('f%d.m4' % ntrack), 'w+b'), do('fr.tofile', V('out')), close('out'),
           digest(T('f%d.m4' % ntrack)),
           open_('fb', 'mark4', T('f%d.m4' % ntrack), 'rb', ntrack=ntrack, decade=2010),
           call('back', 'fb.read_frame'), eq(V('back'), V('fr')), get('back.header.time'), close('fb')]
          for seed, ntrack, fanout, nchan, nsamp in ((21, 64, 4, 8, 80000), (22, 32, 2, 8, 40000))]),

    case('incomplete_and_headerless_streams',
         'a writer given ten samples pads a whole frame as invalid; one frame followed by bare payloads '
         'cannot be opened; two frames followed by payloads open but have no last header '
         '(test_mark4.py, test_incomplete_stream / test_corrupt_stream)',
         open_('fr', 'mark4', M4, 'rs', ntrack=64, decade=2010), call('ten', 'fr.read', 10),
         open_('fw', 'mark4', T('ten.m4'), 'ws', header0=V('fr.header0'), sample_rate=HZ(32e6)),
         do('fw.write', V('ten')), close('fw'), close('fr'), digest(T('ten.m4')),
         [[open_('f', 'mark4', T('ten.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010, fill_value=fv),
           get('f.shape'), call('all', 'f.read'), fn(None, 'allclose_to', V('all'), fv), close('f')]
          for fv in (0.0, -999.0)],
         open_('fb', 'mark4', M4, 'rb', ntrack=64, decade=2010), do('fb.seek', 0xa88),
         call('a', 'fb.read_frame'), call('b', 'fb.read_frame'), close('fb'),
         file_('o1', T('one.m4'), 'w+b'), do('a.tofile', V('o1')), repeat(5, do('a.payload.tofile', V('o1'))), close('o1'),
         open_('s1', 'mark4', T('one.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010),
         file_('o2', T('two.m4'), 'w+b'), do('a.tofile', V('o2')), do('b.tofile', V('o2')),
         repeat(15, do('b.payload.tofile', V('o2'))), close('o2'),
         open_('s2', 'mark4', T('two.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010),
         get('s2.header0'), get('s2._last_header'), close('s2')),

    case('a_frame_missing_in_the_middle',
         'frames 0, 1, 2 and 4 of a sequence: the default reader warns and fills frame 3, verify=True '
         'refuses (test_mark4.py, test_corrupt_stream_missing_frame)',
         open_('fb', 'mark4', M4, 'rb', ntrack=64, decade=2010), do('fb.seek', 0xa88),
         call('a', 'fb.read_frame'), call('b', 'fb.read_frame'), close('fb'),
         fn('dt', 'sub', V('b.header.time'), V('a.header.time')),
         file_('o', T('gap.m4'), 'w+b'), do('a.tofile', V('o')), set_('b.header.mutable', True),
         [[fn('step', 'mul', V('dt'), k, quiet=True), fn('when', 'add', V('a.header.time'), V('step'), quiet=True),
           set_('b.header.time', V('when')), do('b.tofile', V('o'))] for k in (1, 2, 4)],
         close('o'), digest(T('gap.m4')),
         open_('f2', 'mark4', T('gap.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010),
         get('f2.start_time'), get('f2.stop_time'), get('f2.shape'), call(None, 'f2.read'), close('f2'),
         open_('f3', 'mark4', T('gap.m4'), 'rs', sample_rate=HZ(32e6), ntrack=64, decade=2010, verify=True),
         get('f3.stop_time'), call(None, 'f3.read'), close('f3')),

    case('writer_from_keywords',
         'a stream written from keywords whose first frame is the last of a second; the file read back '
         'finds its rate; pickled readers (test_mark4.py, test_start_at_last_frame / test_stream_writer / '
         'test_pickle)',
         open_('fw', 'mark4', T('kw.m4'), 'ws', sample_rate=HZ(32e6), time=TIME('2012-01-01T23:59:59.997500000'), ntrack=32,
               sample_shape=TUP(4), fanout=4, bps=2),
         gets('fw', 'sample_rate', 'samples_per_frame', 'sample_shape', 'start_time'),
         do('fw.write', RNG(31, (160000, 4), LEVELS2)), get('fw.time'), close('fw'), digest(T('kw.m4')),
         open_('fr', 'mark4', T('kw.m4'), 'rs', decade=2010), get('fr.sample_rate'), get('fr.start_time'),
         get('fr.stop_time'), get('fr.shape'), do('fr.seek', 80000 - 3),
         fn('fp', 'pickle_roundtrip', V('fr'), quiet=True), call(None, 'fp.read', 6), close('fp'), close('fr')),
]
