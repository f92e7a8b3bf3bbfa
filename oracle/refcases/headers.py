"""Header arithmetic that needs no samples: days, decades, fractional seconds, names of files."""
from ._dsl import *    # noqa: F401,F403


def mjd(day):
    """ISO date of a whole Modified Julian Day (1858-11-17 + day)."""
    import datetime
    return (datetime.date(1858, 11, 17) + datetime.timedelta(days=day)).isoformat() + 'T00:00:00'


CASES = [
    case('thousands_of_days_and_decades',
         'Mark 5B stores the day modulo 1000, Mark 4 the year modulo 10: both are completed from a '
         'reference time to the nearest possibility (mark5b/tests/test_mark5b.py, test_infer_kday; '
         'mark4/tests/test_mark4.py, test_infer_decade)',
         [[call('h', 'mark5b.Mark5BHeader', None, verify=False, quiet=True), set_('h.jday', jday),
           do('h.infer_kday', TIME(mjd(ref))), get('h.kday')]
          for jday, ref in ((882, 57500), (120, 57500), (882, 57113), (120, 57762), (263, 57762), (261, 57762))],
         [[call('m', 'mark4.Mark4Header', None, ntrack=16, verify=False, quiet=True), setitem('m', 'bcd_unit_year', year),
           do('m.infer_decade', TIME(ref)), get('m.decade')]
          for year, ref in ((5, '2014-01-01T12:00:00'), (5, '2009-12-28T19:27:33'), (4, '2009-01-01T19:27:33'),
                            (3, '2018-04-27T06:42:15'), (4, '2018-04-27T06:42:15'))],
         gpu=False),

    case('guppi_start_time_in_three_cards',
         'a start time a day and a quarter and 2**-10 day later lands in whole days, whole seconds and the '
         'rest; written and read back it is the same header (guppi/tests/test_guppi.py, '
         'test_fractional_time_header)',
         file_('fh', S('sample_puppi.raw'), 'rb'), call('h0', 'guppi.GUPPIHeader.fromfile', V('fh')), close('fh'),
         call('h1', 'h0.copy'), fn('later', 'add', V('h0.start_time'), NS(int((1.25 + 2 ** -10) * 86400 * 10 ** 9)), quiet=True),
         set_('h1.start_time', V('later')), item(None, 'h1', 'STT_IMJD'), item(None, 'h1', 'STT_SMJD'),
         item(None, 'h1', 'STT_OFFS'), get('h1.time'), get('h1.start_time'),
         file_('out', T('h.raw'), 'w+b'), do('h1.tofile', V('out')), do('out.seek', 0),
         call('h2', 'guppi.GUPPIHeader.fromfile', V('out')), close('out'), eq(V('h2'), V('h1')), get('h2.time'),
         call(None, 'guppi.GUPPIHeader.fromvalues', nchan=1, npol=1, bps=4, samples_per_frame=10001),
         call(None, 'dada.DADAHeader.fromvalues', nchan=1, npol=1, complex_data=False, bps=4, samples_per_frame=10001),
         gpu=False),

    case('guppi_leap_second_start_times',
         'a start time inside an inserted leap second, 23:59:60.375, is kept to the nanosecond: the cards '
         'hold the next day, minus one second, plus the fraction; the times around it are ordinary '
         '(guppi/tests/test_guppi.py, test_leap_seconds)',
         [[call('h', 'guppi.GUPPIHeader.fromvalues', start_time=TIME(t), quiet=True), item(None, 'h', 'STT_IMJD'),
           item(None, 'h', 'STT_SMJD'), item(None, 'h', 'STT_OFFS'), get('h.start_time'),
           fn(None, 'sub', V('h.start_time'), TIME(t))]
          + ([fn('later', 'add', V('h.start_time'), NS(1000000000)), fn(None, 'sub', V('later'), TIME(t))] if k < 2 else [])
          # (a second added to 23:59:59.5 is 23:59:60.5 for the reference and the next midnight's
          # 00:00:00.5 on numpy's scale, which this package's header times live on: not compared)
          for k, t in enumerate(('2012-06-30T23:59:60.375000000', '2012-07-01T00:00:00.125000000',
                                 '2012-06-30T23:59:59.500000000'))],
         # (exactly 23:59:60: the reference's cards carry the rounding of its day arithmetic --
         # 86399 s + 0.99999999998 -- so only the instant is compared)
         call('h0', 'guppi.GUPPIHeader.fromvalues', start_time=TIME('2012-06-30T23:59:60.000000000'), quiet=True),
         get('h0.start_time'),
         gpu=False),

    case('file_names_from_headers',
         'name templates filled from header cards: PUPPI scan names, DADA names by frame number and by '
         'byte offset (guppi/tests/test_guppi.py, TestGUPPIFileNameSequencer; dada/tests/test_dada.py, '
         'TestDADAFileNameSequencer)',
         file_('fg', S('sample_puppi.raw'), 'rb'), call('gh', 'guppi.GUPPIHeader.fromfile', V('fg')), close('fg'),
         call('gn', 'guppi.base.GUPPIFileNameSequencer', 'puppi_{stt_imjd}_{src_name}_{scannum}.{file_nr:04d}.raw', V('gh'),
              quiet=True),
         item(None, 'gn', 0), item(None, 'gn', 29),
         call('dn', 'dada.base.DADAFileNameSequencer', '{obs_offset:06d}.x', {'OBS_OFFSET': 10, 'FILE_SIZE': 20},
              quiet=True),
         item(None, 'dn', 0), item(None, 'dn', 9),
         call(None, 'dada.base.DADAFileNameSequencer', '{obs_offset:06d}.x', {'OBS_OFFSET': 10}, quiet=True),
         file_('fd', S('sample.dada'), 'rb'), call('dh', 'dada.DADAHeader.fromfile', V('fd')), close('fd'),
         call('d1', 'dada.base.DADAFileNameSequencer', '{frame_nr}_{obs_offset:016d}.dada', V('dh'), quiet=True),
         item(None, 'd1', 0), item(None, 'd1', 1), item(None, 'd1', 10),
         call('d2', 'dada.base.DADAFileNameSequencer', '{utc_start}_{obs_offset:016d}.000000.dada', V('dh'), quiet=True),
         item(None, 'd2', 0), item(None, 'd2', 100),
         call('vn', 'sf.FileNameSequencer', 'x{file_nr:03d}_{edv}.vdif', {'edv': 3}, quiet=True),
         item(None, 'vn', 0), item(None, 'vn', 12), item(None, 'vn', -1),
         gpu=False),

    case('vdif_header_kinds',
         'the header class follows the EDV; sample.vdif\'s header field by field; headers of one stream '
         'compared; what verification refuses (vdif/tests/test_vdif.py, TestVDIF.test_header)',
         [[file_('f', S(name), 'rb'), call('h', 'vdif.VDIFHeader.fromfile', V('f')), close('f'), get('h'),
           gets('h', 'edv', 'nbytes', 'frame_nbytes', 'payload_nbytes', 'samples_per_frame', 'nchan', 'bps',
                'complex_data', 'station'),
           item(None, 'h', 'ref_epoch'),
           item(None, 'h', 'seconds'), item(None, 'h', 'frame_nr'), item(None, 'h', 'thread_id'),
           item(None, 'h', 'vdif_version'), item(None, 'h', 'legacy_mode'), item(None, 'h', 'invalid_data')]
          for name in ('sample.vdif', 'sample_mwa.vdif', 'sample_arochime.vdif', 'sample_bps1.vdif', 'sample_vlbi.vdif')],
         file_('f', S('sample.vdif'), 'rb'), call('a', 'vdif.VDIFHeader.fromfile', V('f')), do('f.seek', 5032),
         call('b', 'vdif.VDIFHeader.fromfile', V('f')), close('f'),
         call(None, 'a.same_stream', V('b')), eq(V('a'), V('b')), get('a.time'), get('a.sample_rate'),
         get('a.frame_rate'), item(None, 'a', 'sampling_rate'), item(None, 'a', 'sampling_unit'),
         item(None, 'a', 'loif_tuning'), item(None, 'a', 'personality'),
         call('c', 'a.copy'), setitem('c', 'sync_pattern', 0), do('c.verify'),
         call('d', 'a.copy'), setitem('d', 'frame_length', 10), do('d.verify'), get('d.frame_nbytes'),
         call('e', 'a.copy'), set_('e.nchan', 3), set_('e.bps', 32), get('e.bps'), set_('e.bps', 33),
         setitem('a', 'frame_nr', 5),
         gpu=False),

    case('mark4_track_assignments',
         'the header of each sample recording: fan-out, channels, which converter and sideband every '
         'track carries, and the time with its millisecond digits (mark4/tests/test_mark4.py, '
         'TestMark4.test_header and the per-layout header tests)',
         [[open_('fb', 'mark4', S(name), 'rb', ntrack=ntrack, decade=2010), call(None, 'fb.locate_frames'),
           call('h', 'fb.find_header'), close('fb'), get('h'),
           gets('h', 'ntrack', 'fanout', 'nchan', 'bps', 'nsb', 'samples_per_frame', 'frame_nbytes', 'payload_nbytes',
                'nbytes', 'time', 'fraction', 'converters', 'track_assignment'),
           item(None, 'h', 'fan_out'), item(None, 'h', 'magnitude_bit'), item(None, 'h', 'converter_id'),
           item(None, 'h', 'lsb_output'), item(None, 'h', 'bcd_headstack1'), item(None, 'h', 'system_id'),
           item(None, 'h', 'bcd_fraction'), item(None, 'h', 'crc')]
          for name, ntrack in (('sample.m4', 64), ('sample_32track.m4', 32), ('sample_32track_fanout2.m4', 32),
                               ('sample_16track.m4', 16), ('sample_64track_fanout2_ft.m4', 64))],
         gpu=False),
]
