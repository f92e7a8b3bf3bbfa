"""GSB: time stamps in a text file, samples in one (rawdump) or 2 x 2 (phased) raw files."""
from ._dsl import *    # noqa: F401,F403

STREAM_FACTS = ('sample_rate', 'samples_per_frame', 'sample_shape', 'shape', 'size', 'ndim', 'bps', 'complex_data',
                'start_time', 'stop_time', 'time', 'fill_value', 'squeeze', 'subset', 'verify')
TS_RAW, RAW = S('gsb/sample_gsb_rawdump.timestamp'), S('gsb/sample_gsb_rawdump.dat')
TS_PH = S('gsb/sample_gsb_phased.timestamp')
PHASED = [[S('gsb/sample_gsb_phased.Pol-%s%d.dat' % (p, k)) for k in (1, 2)] for p in 'LR']
RATE_RAW = (1e8 / 3) / 2 ** 23 * 2 ** 12 * 2          # 4 bits, 4096-byte payloads
PN_PH = 2 ** 12                                        # phased sample: 4096 bytes per file and frame
RATE_PH = (1e8 / 3) / 2 ** 23 * PN_PH / 512            # 8 bits complex, 512 channels: 8 samples per frame

CASES = [
    case('rawdump_stream',
         'the raw-dump sample: 4-bit real samples, ten frames; shape, times, the last ten samples, '
         'read(out=), the same with samples_per_frame instead of payload_nbytes '
         '(gsb/tests/test_gsb.py, test_raw_stream)',
         open_('fh', 'gsb', TS_RAW, 'rs', raw=RAW, sample_rate=HZ(RATE_RAW), payload_nbytes=4096, squeeze=False),
         gets('fh', *STREAM_FACTS), get('fh.header0'), get('fh._last_header'), get('fh.payload_nbytes'),
         call(None, 'fh.readable'), call(None, 'fh.writable'),
         call(None, 'fh.read', 8192), do('fh.seek', -10, 2), call(None, 'fh.read', 10), get('fh.time'),
         do('fh.seek', 0), call('all', 'fh.read'), call(None, 'fh.tell'),
         do('fh.seek', 0), let('buf', ZEROS((81920, 1))), do('fh.read', out=V('buf')), eq(V('buf'), V('all')),
         do('fh.seek', 1, 'end'), call(None, 'fh.read'), close('fh'),
         open_('f2', 'gsb', TS_RAW, 'rs', raw=RAW, sample_rate=HZ(RATE_RAW), samples_per_frame=8192),
         get('f2.sample_shape'), get('f2.shape'), get('f2.payload_nbytes'), call('flat', 'f2.read'), close('f2'),
         open_('fw', 'gsb', T('t.timestamp'), 'ws', raw=T('t.dat'), header0=V('fh.header0'), sample_rate=HZ(RATE_RAW),
               samples_per_frame=8192),
         get('fw.sample_rate'), do('fw.write', V('flat')), close('fw'),
         digest(T('t.dat')), digest(RAW), digest(T('t.timestamp')), digest(TS_RAW),
         open_('fn', 'gsb', T('t.timestamp'), 'rs', raw=T('t.dat'), sample_rate=HZ(RATE_RAW), samples_per_frame=8192),
         get('fn.header0'), get('fn._last_header'), get('fn.stop_time'), call('again', 'fn.read'),
         eq(V('again'), V('flat')), close('fn')),

    case('phased_stream',
         'the phased sample: 8-bit complex, 512 channels, two polarisations in two files each; frames '
         'interleave the files; subsets; writing back gives the four files again '
         '(test_gsb.py, test_phased_stream)',
         open_('fh', 'gsb', TS_PH, 'rs', raw=PHASED, sample_rate=HZ(RATE_PH), payload_nbytes=PN_PH, squeeze=False),
         gets('fh', *STREAM_FACTS), get('fh.header0'), get('fh._last_header'),
         call(None, 'fh.read', 8), do('fh.seek', -8, 2), call(None, 'fh.read', 8), get('fh.time'),
         do('fh.seek', 0), call('all', 'fh.read'), call(None, 'fh.tell'),
         do('fh.seek', 1, 'end'), call(None, 'fh.read'), close('fh'),
         open_('f2', 'gsb', TS_PH, 'rs', raw=PHASED, sample_rate=HZ(RATE_PH), samples_per_frame=8,
               subset=TUP(1, SL(0, 512, 64))),
         get('f2.sample_shape'), get('f2.shape'), call(None, 'f2.read', 20), close('f2'),
         open_('fw', 'gsb', T('p.timestamp'), 'ws',
               raw=[[T('pL1.dat'), T('pL2.dat')], [T('pR1.dat'), T('pR2.dat')]], header0=V('fh.header0'),
               sample_rate=HZ(RATE_PH), samples_per_frame=8),
         do('fw.write', V('all')), close('fw'),
         [[digest(T('p%s%d.dat' % (p, k))), digest(S('gsb/sample_gsb_phased.Pol-%s%d.dat' % (p, k)))]
          for p in 'LR' for k in (1, 2)],
         digest(T('p.timestamp')), digest(TS_PH)),

    case('one_file_per_polarisation',
         'only the first file of each polarisation, and a single polarisation: the frames hold half the '
         'samples (test_gsb.py, test_phased_stream_one_file_per_pol)',
         open_('fh', 'gsb', TS_PH, 'rs', raw=[[PHASED[0][0]], [PHASED[1][0]]], sample_rate=HZ(RATE_PH / 2),
               payload_nbytes=PN_PH),
         get('fh.sample_shape'), get('fh.shape'), get('fh.samples_per_frame'), call(None, 'fh.read', 6),
         do('fh.seek', -2, 2), call(None, 'fh.read'), close('fh'),
         open_('f1', 'gsb', TS_PH, 'rs', raw=[[PHASED[0][0], PHASED[0][1]]], sample_rate=HZ(RATE_PH),
               payload_nbytes=PN_PH),
         get('f1.sample_shape'), get('f1.shape'), call(None, 'f1.read', 3), close('f1')),

    case('time_stamp_lines',
         'headers parsed from time-stamp lines of both kinds, their times, and the lines written back '
         '(test_gsb.py, TestGSB.test_header / test_header_seek_offset / timestamp io)',
         [[file_('ft', ts, 'rt'), call('line', 'ft.readline', quiet=True), do('ft.seek', 0),
           call('h', 'gsb.GSBHeader.fromfile', V('ft')), get('h.mode'), get('h.time'), get('h.nbytes'),
           call(None, 'h.seek_offset', 9), call(None, 'ft.tell'),
           do('ft.seek', V('h.nbytes')), call('h2', 'gsb.GSBHeader.fromfile', V('ft')), get('h2.time'),
           fn(None, 'sub', V('h2.time'), V('h.time')), close('ft'),
           file_('fo', T('line_%d.txt' % k), 'wt'), do('h.tofile', V('fo')), do('h2.tofile', V('fo')), close('fo'),
           digest(T('line_%d.txt' % k)),
           open_('tr', 'gsb', ts, 'rt'), call('h3', 'tr.read_timestamp'), eq(V('h3'), V('h')),
           call(None, 'tr.tell'), call(None, 'tr.get_frame_rate'), close('tr')]
          for k, ts in enumerate((TS_RAW, TS_PH))],
         call('hr', 'gsb.GSBHeader.fromvalues', mode='rawdump', time=TIME('2015-04-27T13:15:00')), get('hr.mode'),
         get('hr.time'), get('hr.nbytes'),
         call('hp', 'gsb.GSBHeader.fromvalues', mode='phased', time=TIME('2013-07-27T21:23:55.3241088'),
              pc_time=TIME('2013-07-27T21:23:55.5'), seq_nr=9, mem_block=3),
         get('hp.mode'), get('hp.time'), get('hp.pc_time'), item(None, 'hp', 'seq_nr'), item(None, 'hp', 'mem_block'),
         item(None, 'hp', 'gps'),
         gpu=False),

    case('payloads_and_frames',
         'payloads of 4-bit real and 8-bit complex data: words from data, data from words; a phased '
         'frame read from the four raw files (test_gsb.py, TestGSB.test_payload / test_frame)',
         let('d4', RNG(41, (512, 1), [-8.0, -3.0, -1.0, 0.0, 2.0, 7.0])),
         call('p4', 'gsb.GSBPayload.fromdata', V('d4'), bps=4), get('p4'), get('p4.data'), item(None, 'p4', SL(5, 9)),
         let('d8', RNG(42, (16, 2, 8), [-128.0, -5.0, 0.0, 3.0, 127.0], complex=True)),
         call('p8', 'gsb.GSBPayload.fromdata', V('d8'), bps=8), get('p8'), get('p8.shape'), get('p8.data'),
         item(None, 'p8', TUP(3, 1, SL(2, 5))),
         file_('ft', TS_RAW, 'rt'), file_('fr', RAW, 'rb'),
         call('f1', 'gsb.GSBFrame.fromfile', V('ft'), V('fr'), bps=4, payload_nbytes=4096), get('f1.header'),
         get('f1.shape'), item(None, 'f1', SL(0, 6)), close('ft'), close('fr'),
         file_('fo_t', T('f.timestamp'), 'wt'), file_('fo_r', T('f.dat'), 'wb'),
         do('f1.tofile', V('fo_t'), V('fo_r')), close('fo_t'), close('fo_r'), digest(T('f.dat')),
         digest(T('f.timestamp'))),

    case('writer_arguments',
         'what the writer refuses: no raw files, one-file phased data with a wrong sample shape; a last '
         'time stamp without samples (test_gsb.py, test_stream_invalid / test_phased_write_one_file)',
         open_('a', 'gsb', T('x.timestamp'), 'ws', sample_rate=HZ(RATE_RAW), samples_per_frame=8192),
         open_('b', 'gsb', TS_RAW, 'rs', sample_rate=HZ(RATE_RAW), samples_per_frame=8192),
         open_('c', 'gsb', TS_RAW, 's', raw=RAW),
         fn('ts', 'file_bytes', TS_RAW, quiet=True), fn('n', 'len', V('ts'), quiet=True),
         [[fn('upto', 'sub', V('n'), 4, quiet=True), fn('part', 'file_bytes', TS_RAW, 0, V('upto'), quiet=True),
           fn(None, 'write_file', T('short%d.timestamp' % k), [V('part')] + extra),
           open_('d', 'gsb', T('short%d.timestamp' % k), 'rs', raw=RAW, payload_nbytes=4096, squeeze=False),
           get('d._last_header'), get('d.shape'), item(None, 'd.info.warnings', 'number_of_frames', prefix=26), close('d')]
          for k, extra in ((0, []), (1, [HEX('78787878')]))],
         fn('line1', 'file_bytes', TS_RAW, 0, 45, quiet=True), fn(None, 'write_file', T('one.timestamp'), [V('line1')]),
         open_('e', 'gsb', T('one.timestamp'), 'rs', raw=RAW, payload_nbytes=4096, squeeze=False),
         get('e._last_header'), get('e.shape'), close('e'),
         open_('f', 'gsb', TS_RAW, 'rs', raw=RAW), get('f.sample_rate'), get('f.samples_per_frame'),
         get('f.payload_nbytes'), close('f'),
         open_('g', 'gsb', TS_PH, 'rs', raw=PHASED, payload_nbytes=32, samples_per_frame=400)),
]
