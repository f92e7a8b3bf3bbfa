"""The format-detecting entry points (open, file_info), info objects, sequences of files."""
from ._dsl import *    # noqa: F401,F403

INFO_FACTS = ('format', 'readable', 'missing', 'errors', 'warnings', 'checks')
REF = TIME('2014-01-01T00:00:00')

CASES = [
    case('older_files_of_a_sequence_are_emptied',
         'files left over from an earlier, longer run of the same names do not keep their tails when a '
         'sequence is written over them (round 5 found tails kept by positional writes)',
         [[fn(None, 'write_file', T('s%d.vdif' % k), [FILL(0xee, 50000)])] for k in range(3)],
         open_('fr', 'vdif', S('sample.vdif'), 'rs'), call('d', 'fr.read'),
         open_('fw', 'vdif', [T('s0.vdif'), T('s1.vdif'), T('s2.vdif')], 'ws', header0=V('fr.header0'), nthread=8,
               file_size=8 * 5032),
         do('fw.write', V('d')), close('fw'), close('fr'),
         digest(T('s0.vdif')), digest(T('s1.vdif')), digest(T('s2.vdif')),
         open_('fn', 'vdif', [T('s0.vdif'), T('s1.vdif')], 'rs'), get('fn.shape'), call(None, 'fn.read', 5), close('fn')),

    case('open_finds_the_format',
         'open() without a format: the format is found from the bytes; arguments other formats need are '
         'dropped; squeeze and verify are passed on (baseband/tests/test_core.py, test_open / '
         'test_open_squeeze / test_open_verify)',
         [[call('inf', 'top.file_info', S(name), fmt, nchan=8, ref_time=REF, sample_rate=HZ(32e6)),
           get('inf.format'), get('inf.start_time'),
           open_('fh', 'top', S(name), 'rs', nchan=8, ref_time=REF, sample_rate=HZ(32e6)),
           get('fh.start_time'), get('fh.info.format'), get('fh.shape'), close('fh')]
          for name, fmt in (('sample.m4', 'mark4'), ('sample.m5b', 'mark5b'), ('sample.vdif', 'vdif'))],
         open_('a', 'top', S('sample.vdif'), 'rs', squeeze=False), get('a.sample_shape'), close('a'),
         [[open_('b', 'top', S('sample.vdif'), 'rs', verify=v), get('b.verify'), close('b')] for v in (True, False, 'fix')],
         open_('c', 'top', S('sample.dada')), get('c.info.format'), close('c'),
         open_('d', 'top', S('sample_puppi.raw')), get('d.info.format'), close('d')),

    case('open_refuses',
         'missing arguments are named; wrong ones are found inconsistent with the file; unknown ones are '
         'unexpected; a list of formats narrows the search (test_core.py, test_open_missing_args / '
         'test_open_wrong_args / test_unsupported_file / test_format_with_multiple_formats)',
         open_('a', 'top', S('sample.m4'), 'rs', msg=False), open_('b', 'top', S('sample.m5b'), 'rs'),
         open_('c', 'top', S('sample.m4'), 'rs', sample_rate=HZ(31e6), nchan=8, ref_time=REF),
         open_('d', 'top', S('sample.m4'), 'rs', life=42, nchan=8, ref_time=REF),
         open_('e', 'top', S('sample.vdif'), 'rs', decade=2000),
         open_('f', 'top', S('sample.vdif'), 'rs', kday=55000),
         open_('g', 'top', S('sample.vdif'), 'rs', ref_time=TIME('2000-01-01T12:00:00')),
         open_('h', 'top', S('sample.dada'), 'rs', nchan=8),
         open_('i', 'top', S('sample.m4'), 'rs', decade='2010'),
         open_('j', 'top', S('sample.m5b'), 'rs', kday='unknown', nchan=8, bps=2),
         open_('k', 'top', 'a.a', 'wb', format=TUP('dada', 'mark4')),
         fn(None, 'write_file', T('x.unsupported'), [HEX('6162636465666768696a6b6c6d6e6f707172737475767778797a')]),
         open_('l', 'top', T('x.unsupported')),
         open_('m', 'top', S('sample.vdif'), format=TUP('vdif', 'mark5b')), get('m.info.format'), close('m'),
         open_('n', 'top', S('sample.m4'), format=TUP('vdif', 'mark5b')),
         gpu=False),

    case('file_info_answers',
         'file_info for each sample: what is missing, what was used, whether the file is readable; wrong '
         'types and values of arguments (baseband/tests/test_file_info.py, test_basic_file_info / '
         'test_info_missing_args / test_info_wrong_type_args / test_info_wrong_value_args / test_file_info)',
         [[call('inf', 'top.file_info', S(name)), gets('inf', 'format', 'readable', 'missing', 'errors')]
          for name in ('sample.m4', 'sample.m5b')],
         [[call('inf', 'top.file_info', S(name)), gets('inf', 'format', 'readable', 'errors', 'start_time', 'shape')]
          for name in ('sample.vdif', 'sample.dada', 'sample_puppi.raw')],
         [[call('inf', 'top.file_info', S(name)), gets('inf', 'format', 'readable', 'missing')]
          for name in ('sample_mwa.vdif', 'sample_arochime.vdif')],
         call('i4', 'top.file_info', S('sample.m4'), ref_time=REF, nchan=8, kday=56000),
         gets('i4', 'format', 'readable', 'used_kwargs', 'consistent_kwargs', 'inconsistent_kwargs', 'irrelevant_kwargs',
              'start_time', 'sample_rate', 'shape'),
         call('i5', 'top.file_info', S('sample.m5b'), ref_time=REF, nchan=8, decade=2010),
         gets('i5', 'format', 'readable', 'used_kwargs', 'consistent_kwargs', 'inconsistent_kwargs', 'irrelevant_kwargs',
              'start_time', 'sample_rate', 'shape'),
         call('iv', 'top.file_info', S('sample.vdif'), ref_time=REF, nchan=8, decade=2000),
         gets('iv', 'format', 'used_kwargs', 'consistent_kwargs', 'inconsistent_kwargs', 'irrelevant_kwargs'),
         call('w1', 'top.file_info', S('sample.m4'), decade='2010'), gets('w1', 'format', 'readable'),
         fn(None, 'truth', V('w1.errors')),
         call('w2', 'top.file_info', S('sample.m4'), decade=20100), gets('w2', 'format', 'readable'),
         call('w3', 'top.file_info', S('sample.m5b'), nchan=8, kday=5600000), gets('w3', 'format', 'readable'),
         call('m1', 'top.file_info', S('sample_mwa.vdif'), sample_rate=HZ(1.28e6)),
         gets('m1', 'format', 'readable', 'used_kwargs', 'sample_rate', 'stop_time'),
         fn(None, 'write_file', T('x.unsupported'), [HEX('6162636465666768696a6b6c6d6e6f707172737475767778797a')]),
         call('un', 'top.file_info', T('x.unsupported')), fn(None, 'truth', V('un'))),

    case('info_of_readers',
         'the info of binary and stream readers of each format, called to a dictionary '
         '(the info tests of every format: test_vdif.py / test_mark4.py / test_mark5b.py / test_dada.py / '
         'test_guppi.py / test_gsb.py, *info*)',
         [[open_('fb', fmt, S(name), 'rb', **bkw), call(None, 'fb.info'), close('fb'),
           open_('fs', fmt, S(name), 'rs', **skw), call(None, 'fs.info'), close('fs')]
          for fmt, name, bkw, skw in (
              ('vdif', 'sample.vdif', {}, {}),
              ('vdif', 'sample_mwa.vdif', {}, dict(sample_rate=HZ(1.28e6))),
              ('mark5b', 'sample.m5b', dict(kday=56000, nchan=8), dict(kday=56000, nchan=8)),
              ('mark5b', 'sample.m5b', dict(nchan=8), dict(ref_time=REF, nchan=8, bps=2, sample_rate=HZ(32e6))),
              ('mark4', 'sample.m4', dict(decade=2010), dict(decade=2010)),
              ('mark4', 'sample.m4', dict(ntrack=64), dict(ntrack=64, ref_time=REF)),
              ('dada', 'sample.dada', {}, {}),
              ('guppi', 'sample_puppi.raw', {}, {}))],
         open_('ft', 'gsb', S('gsb/sample_gsb_rawdump.timestamp'), 'rt'), call(None, 'ft.info'), close('ft'),
         open_('fp', 'gsb', S('gsb/sample_gsb_phased.timestamp'), 'rt'), call(None, 'fp.info'), close('fp'),
         open_('gs', 'gsb', S('gsb/sample_gsb_rawdump.timestamp'), 'rs', raw=S('gsb/sample_gsb_rawdump.dat'),
               payload_nbytes=4096),
         call(None, 'gs.info'), close('gs')),

    case('sequences_through_open',
         'lists of names and name sequencers through the format-detecting open, for writing with a named '
         'format and for reading without (test_core.py, test_open_sequence; '
         'helpers/tests/test_sequential_baseband.py)',
         open_('fd', 'top', S('sample.dada')), call('d1', 'fd.read'), call('h1', 'fd.header0.copy'), close('fd'),
         set_('h1.payload_nbytes', 32000),
         open_('fw', 'top', [T('f.0.dada'), T('f.1.dada')], 'ws', format='dada', header0=V('h1')),
         do('fw.write', V('d1')), close('fw'), digest(T('f.0.dada')), digest(T('f.1.dada')),
         open_('fn', 'top', [T('f.0.dada'), T('f.1.dada')]), get('fn.info.format'), fn(None, 'len', V('fn.fh_raw.files')),
         call('again', 'fn.read'), eq(V('again'), V('d1')), close('fn'),
         call('seq', 'sf.FileNameSequencer', T('f{file_nr:03d}.vdif'), quiet=True),
         open_('fv', 'top', S('sample.vdif')), call('d2', 'fv.read'), get('fv.header0', as_='h2', quiet=True),
         open_('fw2', 'vdif', V('seq'), 'ws', header0=V('h2'), nthread=8, file_size=8 * 5032),
         do('fw2.write', V('d2')), close('fw2'), listdir(), digest(T('f001.vdif')),
         open_('fn2', 'top', V('seq')), get('fn2.info.format'), fn(None, 'len', V('fn2.fh_raw.files')),
         call('again2', 'fn2.read'), eq(V('again2'), V('d2')), close('fn2'), close('fv')),

    case('byte_level_sequences',
         'the plain file sequences underneath: sizes, seeking across boundaries, reads that span files, '
         'memmap of a piece; writing with a file size (helpers/tests/test_sequentialfile.py)',
         fn('all', 'file_bytes', S('sample.vdif'), quiet=True),
         [[fn('piece', 'file_bytes', S('sample.vdif'), lo, hi, quiet=True),
           fn(None, 'write_file', T('b%d.bin' % k), [V('piece')])]
          for k, (lo, hi) in enumerate(((0, 10000), (10000, 40001), (40001, 80512)))],
         call('fh', 'sf.open', [T('b0.bin'), T('b1.bin'), T('b2.bin')], 'rb', quiet=True),
         call(None, 'fh.tell'), get('fh.file_nr'), call(None, 'fh.seek', 0, 2), call(None, 'fh.seek', 9990),
         call(None, 'fh.read', 20), call(None, 'fh.tell'), get('fh.file_nr'),
         call(None, 'fh.seek', 39990), call(None, 'fh.read', 100), get('fh.file_nr'),
         call(None, 'fh.seek', -10, 2), call(None, 'fh.read', 100), call(None, 'fh.read', 1),
         call(None, 'fh.seek', 5, 0), call(None, 'fh.seek', 10, 1), call(None, 'fh.seek', -1, 0),
         call('mm', 'fh.memmap', quiet=True, dtype='u1', offset=20000, shape=TUP(16)), fn(None, 'host', V('mm')),
         call(None, 'fh.memmap', dtype='u1', offset=9990, shape=TUP(20)),
         close('fh'),
         call('names', 'sf.FileNameSequencer', T('w{file_nr:02d}.bin'), quiet=True),
         call('fw', 'sf.open', V('names'), 'w+b', file_size=30000, quiet=True),
         do('fw.write', V('all')), call(None, 'fw.tell'), close('fw'), listdir(),
         digest(T('w02.bin')),
         gpu=False),
]
