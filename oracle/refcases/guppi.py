"""GUPPI / PUPPI raw: card headers, channel-first and time-first payloads, overlapping blocks."""
from ._dsl import *    # noqa: F401,F403

STREAM_FACTS = ('sample_rate', 'samples_per_frame', 'sample_shape', 'shape', 'size', 'ndim', 'bps', 'complex_data',
                'start_time', 'stop_time', 'time', 'fill_value', 'squeeze', 'subset', 'verify')
HEADER_FACTS = ('nbytes', 'payload_nbytes', 'frame_nbytes', 'bps', 'complex_data', 'sample_shape', 'samples_per_frame',
                'sample_rate', 'overlap', 'channels_first', 'offset', 'start_time', 'time', 'npol', 'nchan')
P = S('sample_puppi.raw')


def header_without_overlap(name='hw'):
    """Steps: the sample's header with OVERLAP 0 and the payload shortened accordingly."""
    return [open_('fb_', 'guppi', P, 'rb'), call('h_', 'fb_.read_header', quiet=True), close('fb_'),
            call(name, 'h_.copy', quiet=True), set_(name + '.overlap', 0),
            set_(name + '.payload_nbytes', 16384 - 1024)]


CASES = [
    case('stream_with_overlap',
         'four blocks overlapping by 64 samples: the stream drops the overlap except after the last '
         'block; positions near the end; info (guppi/tests/test_guppi.py, test_filestreamer / '
         'test_stream_overlap)',
         open_('fh', 'guppi', P, 'rs'),
         gets('fh', *STREAM_FACTS), get('fh.sample_shape.npol'), get('fh.sample_shape.nchan'), get('fh.header0'), get('fh._last_header'), get('fh.dtype'),
         call(None, 'fh.read', 5), do('fh.seek', 955), call(None, 'fh.read', 10),
         do('fh.seek', 3840), call('tail', 'fh.read'), fn(None, 'len', V('tail')),
         do('fh.seek', -1, 2), call(None, 'fh.tell'), call(None, 'fh.read'), call(None, 'fh.read', 1),
         get('fh.info.format'), get('fh.info.shape'), get('fh.info.sample_rate'), get('fh.info.start_time'),
         get('fh.info.stop_time'), close('fh'),
         open_('f2', 'guppi', P, 'rs', subset=TUP(0, [1, 3])), get('f2.sample_shape'), call(None, 'f2.read', 4), close('f2'),
         open_('f3', 'guppi', P, 'rs', squeeze=False, subset=TUP(1)), get('f3.sample_shape'),
         call(None, 'f3.read', 4), close('f3')),

    case('header_payload_frame',
         'header cards and what follows from them; the first block as a frame, plain and memory-mapped; '
         'payload and frame rebuilt from data write the same bytes (test_guppi.py, TestGUPPI.test_header '
         '/ test_payload / test_frame / test_filereader)',
         open_('fb', 'guppi', P, 'rb'), call('h', 'fb.read_header'), call(None, 'fb.tell'),
         gets('h', *HEADER_FACTS), get('h'),
         do('fb.seek', 0), call('fr', 'fb.read_frame', memmap=False), call(None, 'fb.tell'), get('fr.shape'),
         get('fr.dtype'), get('fr.valid'), item(None, 'fr', SL(0, 3)), item(None, 'fr', TUP(SL(500, 502), 1, SL(1, 3))),
         get('fr.payload'),
         do('fb.seek', 0), call('fm', 'fb.read_frame', memmap=True), item(None, 'fm', SL(0, 3)),
         eq(V('fm.payload'), V('fr.payload')),
         call('fr2', 'fb.read_frame', memmap=False), get('fr2.header.time'), item(None, 'fr2.header', 'PKTIDX'),
         close('fb'),
         call('p', 'guppi.GUPPIPayload.fromdata', V('fr.data'), V('h')), eq(V('p'), V('fr.payload')),
         call('g', 'guppi.GUPPIFrame.fromdata', V('fr.data'), V('h')), eq(V('g'), V('fr')),
         file_('out', T('one.raw'), 'w+b'), do('g.tofile', V('out')), close('out'), digest(T('one.raw')),
         call('hc', 'h.copy'), set_('hc.samples_per_frame', 512), get('hc.payload_nbytes'), item(None, 'hc', 'BLOCSIZE'),
         set_('hc.sample_rate', HZ(500.0)), item(None, 'hc', 'TBIN'),
         set_('hc.time', TIME('2018-01-14T14:11:33.5')), item(None, 'hc', 'STT_IMJD'), item(None, 'hc', 'STT_SMJD'),
         item(None, 'hc', 'STT_OFFS'), item(None, 'hc', 'PKTIDX'), get('hc.time'),
         setitem('h', 'NPOL', 2)),

    case('time_first_blocks',
         'the same samples stored time-first (PKTFMT SIMPLE) with no overlap and 960 samples per block: '
         'bytes written and samples read back (test_guppi.py, test_chan_ordered_stream)',
         open_('fr', 'guppi', P, 'rs'), call('d', 'fr.read', 3840),
         call('h', 'fr.header0.copy'), set_('h.channels_first', False), setitem('h', 'OVERLAP', 0),
         set_('h.samples_per_frame', 960), item(None, 'h', 'PKTFMT'), get('h.payload_nbytes'),
         open_('fw', 'guppi', T('tf.raw'), 'ws', header0=V('h')), do('fw.write', V('d')), close('fw'),
         digest(T('tf.raw')),
         open_('fn', 'guppi', T('tf.raw'), 'rs'), get('fn.shape'), do('fn.seek', 1231), call(None, 'fn.read', 47),
         close('fn'), close('fr')),

    case('file_cut_short',
         'the sample cut inside the last payload, and inside the last header: three blocks remain '
         '(test_guppi.py, test_partial_last_frame)',
         fn('raw', 'file_bytes', P, quiet=True), fn('n', 'len', V('raw')),
         [[fn('upto', 'sub', V('n'), cut, quiet=True), fn('part', 'file_bytes', P, 0, V('upto'), quiet=True),
           fn(None, 'write_file', T('cut%d.raw' % cut), [V('part')]),
           open_('fc', 'guppi', T('cut%d.raw' % cut), 'rs'), get('fc.shape'), get('fc.stop_time'),
           get('fc._last_header'), do('fc.seek', -5, 2), call(None, 'fc.read'), close('fc')]
          for cut in (6091, 17605)]),

    case('several_files',
         'two files of two blocks each, written through a list of names and through a sequentialfile '
         'handle; read back singly, together, and after pickling (test_guppi.py, test_multiple_files_stream)',
         header_without_overlap('hw'),
         open_('fr', 'guppi', P, 'rs'), call('d', 'fr.read', 3840), close('fr'),
         open_('fw', 'guppi', [T('g1.raw'), T('g2.raw')], 'ws', header0=V('hw'), frames_per_file=2),
         get('fw.start_time'), item('a', 'd', SL(None, 1000), quiet=True), do('fw.write', V('a')), get('fw.time'),
         item('b', 'd', SL(1000, None), quiet=True), do('fw.write', V('b')), get('fw.time'), close('fw'),
         digest(T('g1.raw')), digest(T('g2.raw')),
         open_('f2', 'guppi', T('g2.raw'), 'rs'), get('f2.time'), get('f2.shape'), call(None, 'f2.read'), close('f2'),
         open_('fa', 'guppi', [T('g1.raw'), T('g2.raw')], 'rs'), get('fa.start_time'), get('fa.stop_time'),
         call('all', 'fa.read'), eq(V('all'), V('d')), get('fa.time'),
         fn('fp', 'pickle_roundtrip', V('fa'), quiet=True), call(None, 'fp.tell'), do('fp.seek', -10, 2),
         call(None, 'fp.read'), close('fp'), close('fa'),
         fn('fsz', 'mul', V('hw.frame_nbytes'), 2),
         call('sq', 'sf.open', [T('s1.raw'), T('s2.raw')], 'w+b', file_size=V('fsz'), quiet=True),
         open_('fw2', 'guppi', V('sq'), 'ws', header0=V('hw')), do('fw2.write', V('d')), close('fw2'),
         digest(T('s1.raw')), digest(T('s2.raw')),
         open_('bad', 'guppi', [T('g1.raw'), T('g2.raw')], 'wb')),

    case('templates_and_incomplete_writes',
         'a {file_nr} template with frames_per_file, read back through the same template; a writer '
         'closed with ten samples pads the block (test_guppi.py, test_template_stream / '
         'test_incomplete_stream)',
         header_without_overlap('hw'),
         open_('fr', 'guppi', P, 'rs'), call('d', 'fr.read', 3840), close('fr'),
         open_('fw', 'guppi', T('t_{file_nr:04d}.raw'), 'ws', header0=V('hw'), frames_per_file=1),
         do('fw.write', V('d')), close('fw'), listdir(), digest(T('t_0003.raw')),
         open_('ft', 'guppi', T('t_{file_nr:04d}.raw'), 'rs'), get('ft.shape'), do('ft.seek', 955),
         call(None, 'ft.read', 10), close('ft'),
         open_('fi', 'guppi', T('ten.raw'), 'ws', header0=V('hw')), item('ten', 'd', SL(None, 10), quiet=True),
         do('fi.write', V('ten')), close('fi'), digest(T('ten.raw')),
         open_('fj', 'guppi', T('ten.raw'), 'rs'), get('fj.shape'), call(None, 'fj.read', 12), close('fj')),
]
