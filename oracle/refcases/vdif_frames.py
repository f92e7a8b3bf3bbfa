"""VDIF below the stream: binary readers, headers, payloads, frames, frame sets."""
from ._dsl import *    # noqa: F401,F403

HEADER_FACTS = ('edv', 'frame_nbytes', 'payload_nbytes', 'samples_per_frame', 'nchan', 'bps', 'complex_data',
                'station', 'nbytes')
LEVELS2 = [-3.316505, -1.0, 1.0, 3.316505]

CASES = [
    case('binary_reader_walk',
         'read_header / read_frame / read_frameset move the file pointer by whole frames; frame rate and '
         'thread ids come from scanning (test_vdif.py, TestVDIF.test_filereader / test_frameset)',
         open_('fb', 'vdif', S('sample.vdif'), 'rb'),
         call('h', 'fb.read_header'), call(None, 'fb.tell'), gets('h', *HEADER_FACTS), get('h.time'),
         item(None, 'h', 'thread_id'), item(None, 'h', 'frame_nr'), item(None, 'h', 'seconds'),
         do('fb.seek', 0), call('fr', 'fb.read_frame'), call(None, 'fb.tell'),
         get('fr.shape'), get('fr.dtype'), get('fr.sample_shape'), get('fr.valid'), get('fr.nbytes'),
         get('fr.data'), item(None, 'fr', SL(5, 12)), item(None, 'fr', 7),
         do('fb.seek', 0), call('fs', 'fb.read_frameset'), call(None, 'fb.tell'),
         get('fs.shape'), get('fs.sample_shape'), get('fs.nbytes'), get('fs.valid'), get('fs.data'),
         item(None, 'fs', TUP(SL(100, 104), 3)), item(None, 'fs', 'thread_id'),
         call('fs2', 'fb.read_frameset', [3, 5]), get('fs2.shape'), item(None, 'fs2', 'thread_id'), get('fs2.data'),
         do('fb.seek', 0), call(None, 'fb.get_frame_rate'), call(None, 'fb.get_thread_ids'), call(None, 'fb.tell'),
         do('fb.seek', -100, 2), call(None, 'fb.read_frame'),
         close('fb')),

    case('find_header_both_ways',
         'find_header forward / backward from inside a frame, near the end, and with the frame length '
         'taken from a template header (test_vdif.py, test_find_header / test_locate_frames)',
         open_('fb', 'vdif', S('sample.vdif'), 'rb'),
         do('fb.seek', 0), call('h0', 'fb.find_header'), call(None, 'fb.tell'),
         do('fb.seek', 100), call(None, 'fb.find_header'), call(None, 'fb.tell'),
         do('fb.seek', 100), call(None, 'fb.find_header', forward=False), call(None, 'fb.tell'),
         do('fb.seek', -10000, 2), call(None, 'fb.find_header', forward=True), call(None, 'fb.tell'),
         do('fb.seek', -300, 2), call(None, 'fb.find_header', forward=False), call(None, 'fb.tell'),
         do('fb.seek', -300, 2), call(None, 'fb.find_header', forward=True), call(None, 'fb.tell'),
         do('fb.seek', 5032 * 3 + 17), call(None, 'fb.locate_frames', V('h0')),
         do('fb.seek', 5032 * 3 + 17), call(None, 'fb.locate_frames', V('h0'), forward=False),
         do('fb.seek', 5032 * 3 + 17), call(None, 'fb.locate_frames', V('h0'), maximum=100),
         do('fb.seek', 0), call(None, 'fb.locate_frames', V('h0'), check=[-1, 1, 2]),
         do('fb.seek', 40), call(None, 'fb.find_header', V('h0'), maximum=40),
         close('fb')),

    case('find_header_after_a_cut',
         'files whose first frame lost its start: the first whole header is found, for every sample '
         'recording (test_vdif.py, test_find_header_lost_start)',
         [[open_('fb', 'vdif', S(name), 'rb'), call('ha', 'fb.read_header'),
           do('fb.seek', V('ha.frame_nbytes')), call('hb', 'fb.read_header'),
           fn('cutat', 'sub', V('ha.frame_nbytes'), 100, quiet=True), do('fb.seek', V('cutat')),
           call('rest', 'fb.read', quiet=True), close('fb'),
           fn(None, 'write_file', T('cut.vdif'), [V('rest')]),
           open_('fc', 'vdif', T('cut.vdif'), 'rb'), call('hf', 'fc.find_header'), call(None, 'fc.tell'),
           eq(V('hf'), V('hb')), close('fc')]
          for name in ('sample.vdif', 'sample_mwa.vdif', 'sample_arochime.vdif', 'sample_bps1.vdif',
                       'sample_vlbi.vdif')]),

    case('header_words_from_values',
         'fromvalues builds each EDV; time needs a frame rate where the header holds none; legacy headers '
         '(test_vdif.py, TestVDIF.test_header / test_legacy_vdif)',
         call('h3', 'vdif.VDIFHeader.fromvalues', edv=3, time=TIME('2014-06-16T05:56:07.000000000'), samples_per_frame=20000,
              station=65532, sample_rate=HZ(32e6), bps=2, complex_data=False, thread_id=3, nchan=1),
         gets('h3', *HEADER_FACTS), get('h3.time'), get('h3.sample_rate'), get('h3.frame_rate'),
         call('h1', 'vdif.VDIFHeader.fromvalues', edv=1, time=TIME('2010-11-12T13:14:15.25'), sample_rate=HZ(8e6),
              samples_per_frame=16000, station='me', bps=2, complex_data=True, nchan=2),
         gets('h1', *HEADER_FACTS), get('h1.time'), get('h1.sample_rate'),
         call('h0', 'vdif.VDIFHeader.fromvalues', edv=0, time=TIME('2010-11-12T13:14:15.25'), frame_rate=HZ(1600.0),
              samples_per_frame=16000, station='me', bps=2, complex_data=False, nchan=2),
         get('h0'), call(None, 'h0.get_time', frame_rate=HZ(1600.0)), get('h0.time'),
         call('hl', 'vdif.VDIFHeader.fromvalues', edv=False, time=TIME('2010-11-12T13:14:15'), frame_rate=HZ(1600.0),
              samples_per_frame=16000, station='me', bps=2, complex_data=False, nchan=2),
         get('hl'), gets('hl', *HEADER_FACTS),
         file_('out', T('h.bin'), 'w+b'), do('hl.tofile', V('out')), do('h3.tofile', V('out')), close('out'),
         digest(T('h.bin')),
         file_('back', T('h.bin'), 'rb'), call('hl2', 'vdif.VDIFHeader.fromfile', V('back')),
         call('h32', 'vdif.VDIFHeader.fromfile', V('back')), close('back'),
         eq(V('hl2'), V('hl')), eq(V('h32'), V('h3')),
         call(None, 'vdif.VDIFHeader.fromvalues', edv=0x7f, bps=2),
         gpu=False),

    case('payload_encode_decode',
         'payloads from data and back for 2-bit real, 4-bit complex and 8-bit: words, decoded data, item '
         'access (test_vdif.py, TestVDIF.test_payload)',
         [[let('d', RNG(seed, shape, levels, complex=cplx)),
           call('h', 'vdif.VDIFHeader.fromvalues', edv=0, bps=bps, nchan=shape[1], complex_data=cplx,
                samples_per_frame=shape[0], station='aa', time=TIME('2015-01-01T00:00:00'), frame_rate=HZ(100.)),
           call('p', 'vdif.VDIFPayload.fromdata', V('d'), V('h')), get('p'), get('p.shape'), get('p.dtype'),
           get('p.data'), item(None, 'p', SL(3, 9, 2)), item(None, 'p', TUP(5, 1)),
           file_('out', T('p%d.bin' % bps), 'w+b'), do('p.tofile', V('out')), do('out.seek', 0),
           call('p2', 'vdif.VDIFPayload.fromfile', V('out'), V('h')), eq(V('p2'), V('p')), close('out'),
           digest(T('p%d.bin' % bps))]
          for seed, shape, levels, cplx, bps in ((1, (64, 2), LEVELS2, False, 2),
                                                 (2, (32, 4), [-2.0, -1.0, 0.0, 1.0, 2.0], True, 4),
                                                 (3, (16, 2), [-30.0, -3.0, 0.0, 8.0, 100.0], False, 8))]),

    case('frames_built_from_data',
         'a frame from data + header writes the bytes of header and payload; frame sets likewise, with '
         'per-thread headers made from the first (test_vdif.py, TestVDIF.test_frame / test_frameset)',
         let('d', RNG(5, (64, 4, 2), LEVELS2)),
         call('h', 'vdif.VDIFHeader.fromvalues', edv=1, bps=2, nchan=2, complex_data=False, samples_per_frame=64,
              station='bb', time=TIME('2015-01-01T00:00:01'), sample_rate=HZ(4000.)),
         item('d0', 'd', TUP(SL(None), 0), quiet=True),
         call('fr', 'vdif.VDIFFrame.fromdata', V('d0'), V('h')), get('fr'), get('fr.data'),
         call('fs', 'vdif.VDIFFrameSet.fromdata', V('d'), V('h')), get('fs'), get('fs.data'),
         item(None, 'fs', 'thread_id'),
         file_('out', T('fs.vdif'), 'w+b'), do('fs.tofile', V('out')), close('out'), digest(T('fs.vdif')),
         open_('fb', 'vdif', T('fs.vdif'), 'rb'), call('fs2', 'fb.read_frameset'), eq(V('fs2'), V('fs')), close('fb'),
         open_('st', 'vdif', T('fs.vdif'), 'rs'), get('st.shape'), get('st.sample_rate'), call(None, 'st.read'),
         close('st'),
         call('inv', 'vdif.VDIFFrame.fromdata', V('d0'), V('h'), valid=False), get('inv.valid'), get('inv.data'),
         item(None, 'inv', 'invalid_data')),

    case('arochime_partial_copies',
         'copies of the CHIME sample with one thread missing from the second frame set, made frame by '
         'frame: the reader still delivers five sets, the hole filled (test_vdif.py, TestAROCHIMEPartialCopy)',
         open_('fb', 'vdif', S('sample_arochime.vdif'), 'rb'),
         file_('out', T('aro_hole.vdif'), 'w+b'),
         call('f0', 'fb.read_frame'), call('f1', 'fb.read_frame'), call('f2', 'fb.read_frame'),
         call('f3', 'fb.read_frame'), call('rest', 'fb.read', quiet=True), close('fb'),
         do('f0.tofile', V('out')), do('f1.tofile', V('out')), do('f2.tofile', V('out')),
         do('out.write', V('rest')), close('out'), digest(T('aro_hole.vdif')),
         open_('st', 'vdif', T('aro_hole.vdif'), 'rs', sample_rate=HZ(390625.0)),
         get('st.shape'), get('st.start_time'), get('st.stop_time'), call(None, 'st.read'), close('st'),
         # (with verify=False the reference refuses this file -- 'could not find all requested
         # frames' -- where this package still places frames by their headers and fills the
         # hole: a deliberate leniency, not recorded)
         ),

    case('setting_samples_in_place',
         'item assignment on a payload, a frame and a frame set re-encodes the words (test_vdif.py, '
         'test_payload_getitem_setitem / frame and frameset setitem)',
         open_('fb', 'vdif', S('sample.vdif'), 'rb'), call('ro', 'fb.read_frameset'), close('fb'),
         setitem('ro', TUP(SL(0, 3), 2), 1.0),
         call('fs', 'vdif.VDIFFrameSet.fromdata', V('ro.data'), V('ro.header0'), quiet=True), eq(V('fs'), V('ro')),
         item('p', 'fs.frames', 0, quiet=True),
         setitem('p', SL(10, 14), ARRAY([[3.316505], [1.0], [-1.0], [-3.316505]], 'f4')), item(None, 'p', SL(8, 16)),
         get('p.payload'),
         setitem('p', SL(10, 14), ARRAY([3.316505, 1.0, -1.0, -3.316505], 'f4')),
         setitem('fs', TUP(SL(0, 3), 2), ARRAY([[-1.0], [1.0], [-1.0]], 'f4')), item(None, 'fs', TUP(SL(0, 4), 2)),
         setitem('fs', TUP(SL(5, 7)), 1.0), item(None, 'fs', SL(4, 8)),
         get('fs.frames[1].payload'),
         setitem('fs', 'frame_nr', 1234), item(None, 'fs', 'frame_nr'),
         setitem('ro', 'frame_nr', 1234)),
]
