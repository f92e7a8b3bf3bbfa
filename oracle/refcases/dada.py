"""DADA: ASCII headers, one frame per file, MeerKAT beam-former heaps."""
from ._dsl import *    # noqa: F401,F403

STREAM_FACTS = ('sample_rate', 'samples_per_frame', 'sample_shape', 'shape', 'size', 'ndim', 'bps', 'complex_data',
                'start_time', 'stop_time', 'time', 'fill_value', 'squeeze', 'subset', 'verify')
HEADER_FACTS = ('nbytes', 'payload_nbytes', 'frame_nbytes', 'bps', 'complex_data', 'sample_shape', 'samples_per_frame',
                'sample_rate', 'offset', 'start_time', 'time')
D = S('sample.dada')

CASES = [
    case('meerkat_and_beam_former_heaps',
         'MeerKAT headers, and the beam-former recording whose samples are stored in heaps of 256: read, '
         'and written back through the stream writer byte for byte (round 5 found the heap order wrong '
         'in the writer) (test_dada.py, test_meerkat_header / test_meerkat_data / TestMKBF)',
         open_('fm', 'dada', S('sample_meerkat.dada'), 'rs'), gets('fm', *STREAM_FACTS), get('fm.header0'),
         call(None, 'fm.read', 6), close('fm'),
         open_('fk', 'dada', S('sample_mkbf.dada'), 'rs'), gets('fk', *STREAM_FACTS), get('fk.header0'),
         call('all', 'fk.read'), do('fk.seek', 250), call(None, 'fk.read', 6),
         open_('fb', 'dada', S('sample_mkbf.dada'), 'rb'), call('hb', 'fb.read_header'), get('hb.sample_shape'),
         get('hb.payload_nbytes'), call(None, 'fb.read_frame', memmap=False), close('fb'),
         open_('fw', 'dada', T('mkbf.dada'), 'ws', header0=V('fk.header0')), do('fw.write', V('all')), close('fw'),
         digest(T('mkbf.dada')), digest(S('sample_mkbf.dada')),
         call('h3', 'fk.header0.copy'), fn('n3', 'mul', V('h3.payload_nbytes'), 3, quiet=True),
         set_('h3.payload_nbytes', V('n3')), get('h3.samples_per_frame'),
         fn('flip', 'neg', V('all'), quiet=True), item('some', 'all', SL(None, 200), quiet=True),
         item('others', 'flip', SL(200, None), quiet=True),
         open_('fw3', 'dada', T('mkbf3.dada'), 'ws', header0=V('h3')),
         do('fw3.write', V('all')), do('fw3.write', V('some')), do('fw3.write', V('others')), do('fw3.write', V('flip')),
         close('fw3'), digest(T('mkbf3.dada')),
         open_('fr3', 'dada', T('mkbf3.dada'), 'rs'), get('fr3.shape'), do('fr3.seek', 450), call(None, 'fr3.read', 80),
         close('fr3'), close('fk')),

    case('stream_reader_facts',
         'the sample as a stream: facts, first samples, seeking by time and from the end, subset of '
         'polarisations, unsqueezed shape (dada/tests/test_dada.py, test_filestreamer)',
         open_('fh', 'dada', D, 'rs'),
         gets('fh', *STREAM_FACTS), get('fh.header0'), get('fh._last_header'), get('fh.dtype'),
         call('rec', 'fh.read', 12), call(None, 'fh.tell'), get('fh.time', as_='t12'),
         do('fh.seek', 0), do('fh.seek', V('t12')), call(None, 'fh.tell'),
         do('fh.seek', -3, 2), call(None, 'fh.read'), call(None, 'fh.read', 1), close('fh'),
         open_('f2', 'dada', D, 'rs', subset=0), get('f2.sample_shape'), call(None, 'f2.read', 5), close('f2'),
         open_('f3', 'dada', D, 'rs', squeeze=False), get('f3.sample_shape'), get('f3.sample_shape.npol'),
         call(None, 'f3.read', 5), close('f3')),

    case('header_payload_frame',
         'the header read from the file and its derived values; the frame below the stream; payload and '
         'frame rebuilt from data write the same bytes (test_dada.py, TestDADA.test_header / test_payload '
         '/ test_frame / test_filereader)',
         open_('fb', 'dada', D, 'rb'), call('h', 'fb.read_header'), call(None, 'fb.tell'),
         gets('h', *HEADER_FACTS), get('h'),
         do('fb.seek', 0), call('fr', 'fb.read_frame'), call(None, 'fb.tell'), get('fr.shape'), get('fr.dtype'),
         get('fr.valid'), item(None, 'fr', SL(0, 4)), item(None, 'fr', TUP(SL(100, 103), 1)), get('fr.payload'),
         do('fb.seek', 0), call('fm', 'fb.read_frame', memmap=True), item(None, 'fm', SL(0, 4)),
         eq(V('fm.payload'), V('fr.payload')), close('fb'),
         call('p', 'dada.DADAPayload.fromdata', V('fr.data'), V('h')), eq(V('p'), V('fr.payload')),
         call('f2', 'dada.DADAFrame.fromdata', V('fr.data'), V('h')), eq(V('f2'), V('fr')),
         file_('out', T('again.dada'), 'w+b'), do('f2.tofile', V('out')), close('out'),
         digest(T('again.dada')), digest(D),
         call('hk', 'dada.DADAHeader.fromvalues', time=TIME('2013-07-02T01:37:40'), offset=NS(640000), npol=2, bps=8,
              payload_nbytes=64000, sample_rate=HZ(16e6), nchan=1, complex_data=True, instrument='x'),
         gets('hk', *HEADER_FACTS), item(None, 'hk', 'OBS_OFFSET'), item(None, 'hk', 'UTC_START'),
         item(None, 'hk', 'TSAMP'), item(None, 'hk', 'NDIM'),
         call('hc', 'h.copy'), set_('hc.payload_nbytes', 32000), get('hc.samples_per_frame'),
         item(None, 'hc', 'FILE_SIZE'), set_('hc.time', TIME('2013-07-02T01:37:41')), item(None, 'hc', 'OBS_OFFSET'),
         setitem('h', 'NPOL', 1)),

    case('copy_through_a_writer',
         'samples of the reader written with its header give the file back byte for byte; a writer from '
         'keywords (test_dada.py, test_filestreamer writing part)',
         open_('fr', 'dada', D, 'rs'), call('all', 'fr.read'),
         open_('fw', 'dada', T('copy.dada'), 'ws', header0=V('fr.header0')),
         gets('fw', 'sample_rate', 'samples_per_frame', 'sample_shape', 'start_time'),
         do('fw.write', V('all')), get('fw.time'), close('fw'), digest(T('copy.dada')),
         open_('fk', 'dada', T('kw.dada'), 'ws', time=V('fr.start_time'), sample_rate=V('fr.sample_rate'),
               samples_per_frame=16000, npol=2, bps=8, complex_data=True, nchan=1),
         do('fk.write', V('all')), close('fk'),
         open_('bk', 'dada', T('kw.dada'), 'rs'), get('bk.shape'), get('bk.start_time'),
         call('again', 'bk.read'), eq(V('again'), V('all')), close('bk'), close('fr')),

    case('last_frame_shorter',
         'a file cut inside the last frame, and a writer that stops early: the stream ends with the '
         'samples that are there (test_dada.py, test_partial_last_frame)',
         open_('fr', 'dada', D, 'rs'), call('all', 'fr.read'),
         call('h', 'fr.header0.copy'), set_('h.payload_nbytes', 16000),
         open_('fw', 'dada', T('f{frame_nr:d}.dada'), 'ws', header0=V('h')),
         item('most', 'all', SL(None, 14000), quiet=True), do('fw.write', V('most')), close('fw'), listdir(),
         digest(T('f3.dada')),
         open_('fs', 'dada', T('f{frame_nr:d}.dada'), 'rs'), get('fs.shape'), get('fs.stop_time'),
         get('fs._last_header'), do('fs.seek', 11990), call(None, 'fs.read'), close('fs'),
         fn('raw', 'file_bytes', D, 0, 4096 + 64000 - 2000, quiet=True),
         fn(None, 'write_file', T('cut.dada'), [V('raw')]),
         open_('fc', 'dada', T('cut.dada'), 'rs'), get('fc.shape'), get('fc.stop_time'),
         do('fc.seek', -4, 2), call(None, 'fc.read'), close('fc'), close('fr')),

    case('one_frame_per_second',
         'a header whose frame lasts exactly a second (test_dada.py, test_one_frame_per_second)',
         open_('fr', 'dada', D, 'rs'), call('all', 'fr.read'),
         call('h', 'fr.header0.copy'), set_('h.sample_rate', HZ(16000.0)),
         open_('fw', 'dada', T('slow.dada'), 'ws', header0=V('h')), do('fw.write', V('all')), get('fw.time'),
         close('fw'),
         open_('fs', 'dada', T('slow.dada'), 'rs'), get('fs.sample_rate'), get('fs.start_time'), get('fs.stop_time'),
         call('again', 'fs.read'), eq(V('again'), V('all')), close('fs'), close('fr')),

    case('files_named_by_template',
         'the DADA naming convention as a template: files named by time and byte offset, read back by the '
         'same template or by a list of names; a pickled reader (test_dada.py, test_template_stream / '
         'test_multiple_files_stream / test_pickle)',
         open_('fr', 'dada', D, 'rs'), call('all', 'fr.read'),
         call('h', 'fr.header0.copy'), set_('h.payload_nbytes', 16000),
         open_('fw', 'dada', T('{utc_start}_{obs_offset:016d}.000000.dada'), 'ws', header0=V('h')),
         do('fw.write', V('all')), close('fw'), listdir('names'),
         open_('f0', 'dada', T('{utc_start}_{obs_offset:016d}.000000.dada'), 'rs'),
         open_('f1', 'dada', T('{utc_start}_{obs_offset:016d}.000000.dada'), 'rs',
               UTC_START='2013-07-02-01:37:40', OBS_OFFSET=6400000000 + 16000, FILE_SIZE=16000),
         get('f1.shape'), get('f1.start_time'), do('f1.seek', 3995), call(None, 'f1.read', 10),
         fn('fp', 'pickle_roundtrip', V('f1'), quiet=True), call(None, 'fp.tell'), call(None, 'fp.read', 3), close('fp'),
         close('f1'),
         open_('fw2', 'dada', T('a{frame_nr:02d}.dada'), 'ws', header0=V('h')), do('fw2.write', V('all')), close('fw2'),
         open_('f2', 'dada', [T('a00.dada'), T('a01.dada'), T('a02.dada'), T('a03.dada')], 'rs'),
         get('f2.shape'), call('again', 'f2.read'), eq(V('again'), V('all')), close('f2'), close('fr')),

    case('writer_stopped_mid_frame',
         'ten samples into a 16000-sample frame: the writer pads, the file keeps its full size '
         '(test_dada.py, test_incomplete_stream)',
         open_('fr', 'dada', D, 'rs'), call('ten', 'fr.read', 10),
         open_('fw', 'dada', T('ten.dada'), 'ws', header0=V('fr.header0')), do('fw.write', V('ten')), close('fw'),
         close('fr'), digest(T('ten.dada')),
         open_('fs', 'dada', T('ten.dada'), 'rs'), get('fs.shape'), call(None, 'fs.read', 12), close('fs')),
]
